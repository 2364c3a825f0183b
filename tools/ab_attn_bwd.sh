# same-box A/B of the training step (forward + backward, O96, bf16) on two builds of the kernel library: usage: bash tools/ab_attn_bwd.sh "tree bwdold"
R=$PWD; export ANEMOI_TORCH_EXT=0
for rep in 1 2; do for v in $1; do
  if [ $v = tree ]; then unset ANEMOI_HIP_LIB; else export ANEMOI_HIP_LIB=$R/anemoi_core_amd/lib/alt_$v.so; fi
  echo "== $v"; python tools/train_step_time.py 2>/dev/null | tail -3
done; done
