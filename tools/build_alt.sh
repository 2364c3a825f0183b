# usage: bash tools/build_alt.sh NAME [extra hipcc flags] [-- FILE.hip ...]  -> anemoi_core_amd/lib/alt_NAME.so
# A/B builds (ANEMOI_HIP_LIB=... selects one): the named translation units (default: gt_attention.hip) are recompiled with the
# extra flags, every other object is the tree build's (python -m anemoi_core_amd.build first).
set -e
NAME=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
FLAGS=(); FILES=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; FILES=("$@"); break; fi; FLAGS+=("$1"); shift; done
[ ${#FILES[@]} -eq 0 ] && FILES=(gt_attention.hip)
O=/tmp/alt_$NAME; rm -rf $O; mkdir -p $O
cp $R/anemoi_core_amd/lib/obj/*.o $O/
for f in "${FILES[@]}"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I $R/include -I $R/anemoi_core_amd/csrc -w "${FLAGS[@]}" -x hip -c $R/anemoi_core_amd/csrc/$f -o $O/${f%.*}.o ) &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/anemoi_core_amd/lib/alt_$NAME.so $O/*.o
echo built $R/anemoi_core_amd/lib/alt_$NAME.so
