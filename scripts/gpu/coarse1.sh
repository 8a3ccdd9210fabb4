#!/bin/bash
# where do the coarse-level convolution launches spend their time?  scripts/kbench.py on levels 3-4 (the 4th and 5th level of the
# pair batch) with one component of spconv16x_kernel compiled out at a time (timing only, wrong results)
mkdir -p gpurun_out/r04r; O=gpurun_out/r04r; rm -f $O/coarse.txt
for v in base x3_nomfma x3_nogather x3_nodma x3_nosplit x3_nobarrier x3_nobfrag; do
  L=""; [ $v != base ] && L=$PWD/pointcontrast_amd/libpcmi_$v.so
  echo "== $v" >> $O/coarse.txt
  PCMI_LIB=$L KBENCH_LEVELS=2,3,4 KBENCH_SUSTAINED=0 PYTHONPATH=. timeout 120 python scripts/kbench.py 2>/dev/null | grep "3^3" >> $O/coarse.txt
done
cat $O/coarse.txt
