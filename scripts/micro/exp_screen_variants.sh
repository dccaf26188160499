# the timed step with tuning flags of the 16-bit (screening) forward, one after the other on the same box
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
for v in "base" "--ce-fused-attn-out" "--ce-fused-norm-mlp" "base"; do
  a=""; [ "$v" != "base" ] && a="$v"
  timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-extras --cpu-queries 0 $a > gpurun_out/r3d/out.json 2> gpurun_out/r3d/err.log
  python - "$v" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r3d/out.json'))
print(sys.argv[1], d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['rerank_screen']['rescored_fraction'], d['rerank_screen']['margin_widened_in_timed_steps'], d['precision_check'] if 'precision_check' in d else '')
PY
done
