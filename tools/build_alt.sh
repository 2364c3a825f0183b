# usage: bash tools/build_alt.sh NAME [extra hipcc flags]  -> anemoi_core_amd/lib/alt_NAME.so (A/B builds: ANEMOI_HIP_LIB=... selects one)
set -e
NAME=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
O=/tmp/alt_$NAME; mkdir -p $O
for f in lib.cpp gt_attention.hip gt_attention_bwd.hip rowwise.hip rowwise_bwd.hip linear.hip wgrad.hip; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I $R/include -I $R/anemoi_core_amd/csrc -w "$@" -x hip -c $R/anemoi_core_amd/csrc/$f -o $O/${f%.*}.o ) &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/anemoi_core_amd/lib/alt_$NAME.so $O/*.o
echo built $R/anemoi_core_amd/lib/alt_$NAME.so
