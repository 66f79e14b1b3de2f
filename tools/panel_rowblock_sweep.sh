# sweeps of the panel kernel's forms on the row block of one rank (N = 8, 4, 2); results under gpurun_out/
B=tools/bench_spmv_variants.py
o=gpurun_out/rowblock_slices.txt
: > $o
echo "# N=8 block (1.25 M rows): blocked, auto panel (slices), ring" >> $o
timeout -k 10 200 python $B --row1 1250000 --reps 50 blocked panel panel:4883:17:4:7 >> $o 2>&1
echo "# N=8 forced slices 1/2/4/8" >> $o
for f in 1 2 4 8; do SPL_PANEL_SLICES=$f timeout -k 10 200 python $B --row1 1250000 --reps 50 panel >> $o 2>&1; done
echo "# N=4 block" >> $o
timeout -k 10 200 python $B --row1 2500000 --reps 30 blocked panel >> $o 2>&1
SPL_PANEL_SLICES=1 timeout -k 10 200 python $B --row1 2500000 --reps 30 panel >> $o 2>&1
echo "# N=2 block" >> $o
timeout -k 10 200 python $B --row1 5000000 --reps 30 panel >> $o 2>&1
SPL_PANEL_SLICES=8 timeout -k 10 200 python $B --row1 5000000 --reps 30 panel >> $o 2>&1
echo "# N=1" >> $o
timeout -k 10 200 python $B --reps 30 panel >> $o 2>&1
SPL_PANEL_SLICES=8 timeout -k 10 200 python $B --reps 30 panel >> $o 2>&1
grep "spec\|^#" $o | cut -c1-130
