# usage: bash tools/ab_tree.sh OTHER_TREE  -> bench of this tree and of another checkout (e.g. a git worktree with its own build) alternating
for rep in 1 2; do for t in $1 . $1 .; do
  (cd $t && python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("'$t'", round(d["ms_per_step"],4))')
done; done
