set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
for v in "base" "--pipeline" "--ce-fused-attn-norm" "--screen-safety 1.25"; do
  a=""; [ "$v" != "base" ] && a="$v"
  timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-extras --cpu-queries 0 $a > gpurun_out/r3d/out.json 2> gpurun_out/r3d/err.log
  python - "$v" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r3d/out.json'))
print(sys.argv[1], d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['rerank_screen']['rescored_fraction'], d['rerank_screen']['margin_widened_in_timed_steps'])
PY
done
