R=${GRAFT_REPO_ROOT:-/root/repo}
for o in "warp_multi_lds=1" "warp_multi_lds=0" "warp_multi_lds=2" "warp_multi_lds=1" "warp_multi_lds=0"; do
  FFWM_OPTS=$o python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --no-kernels 2>/dev/null > /tmp/o.txt
  python - "$o" <<'P'
import json,sys
L=open("/tmp/o.txt").read().strip().splitlines()
k=json.loads(L[0])["kernels"]; d=json.loads(L[-1])
row=[r for r in k if r["kernel"]=="warp_flipcat_fwd_multi"]
print(sys.argv[1], d["ms_per_step"], "ms;", "fwd_multi", row[0]["avg_us"] if row else None, "us; roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
P
done
