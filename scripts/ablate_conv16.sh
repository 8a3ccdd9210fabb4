#!/bin/bash
# Builds libpcmi_abl{1..5}.so: the product library with spconv.hip compiled under -DPCMI_ABLATE=n (timing ablations of
# spconv16_kernel, see spconv.hip).  Run here (hipcc cross-compiles); on the GPU box:
#   for n in 0 1 2 3 4 5; do PCMI_LIB=pointcontrast_amd/libpcmi_abl$n.so KBENCH_LEVELS=0 KBENCH_SUSTAINED=0 python scripts/kbench.py; done
set -e
cd "$(dirname "$0")/.."
python -m pointcontrast_amd.build >/dev/null
B=pointcontrast_amd/csrc/_build
for n in ${ABLATIONS:-1 2 3 4 5}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -Wno-unused-function -DPCMI_ABLATE=$n \
    -c pointcontrast_amd/csrc/spconv.hip -o $B/spconv_abl$n.o &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -Wno-unused-function -DPCMI_ABLATE=$n \
    -c pointcontrast_amd/csrc/spconv_wgrad.hip -o $B/spconv_wgrad_abl$n.o &
done
wait
for n in ${ABLATIONS:-1 2 3 4 5}; do
  objs=$(ls $B/*.o | grep -v "spconv\.o" | grep -v "spconv_wgrad\.o" | grep -v "_abl")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o pointcontrast_amd/libpcmi_abl$n.so $objs $B/spconv_abl$n.o $B/spconv_wgrad_abl$n.o
done
cp pointcontrast_amd/libpcmi.so pointcontrast_amd/libpcmi_abl0.so
ls -la pointcontrast_amd/libpcmi_abl*.so
