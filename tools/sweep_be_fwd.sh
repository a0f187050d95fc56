for opts in "xcd_remap=0 rows_per_thread=1" "xcd_remap=0 rows_per_thread=2" "xcd_remap=0 rows_per_thread=4" "xcd_remap=1 rows_per_thread=4" "xcd_remap=0 rows_per_thread=4 ablate=1" "xcd_remap=0 rows_per_thread=4 ablate=2" "xcd_remap=0 rows_per_thread=4 channel_slab=8" "xcd_remap=0 rows_per_thread=4 channel_slab=32" "xcd_remap=1 rows_per_thread=4 channel_slab=32" "xcd_remap=0 rows_per_thread=2 channel_slab=32"; do
  args=""; for o in $opts; do args="$args --opt $o"; done
  echo "== $opts"; timeout 100 python tools/kbench.py --reps 10 --only be_fwd $args 2>&1 | grep avg_ms | python -c "import sys,json; [print(json.loads(l)['avg_ms'], json.loads(l)['GBps']) for l in sys.stdin]"
done
