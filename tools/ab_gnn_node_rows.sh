# GNN forward (ms, 40 graph replays) against the node chain's rows per panel: usage: bash tools/ab_gnn_node_rows.sh "0 48 44 36"
export ANEMOI_TORCH_EXT=0
for rep in 1 2; do for b in $1; do
  if [ $b = 0 ]; then unset ANEMOI_GNN_NODE_ROWS; else export ANEMOI_GNN_NODE_ROWS=$b; fi
  python bench.py --config gnn --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("forward gnn node_rows '$b'", round(d["ms_per_step"],4))'
done; done
