for t in 65536 131072 262144 524288; do
  RAG_CE_TOKENS=$t timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-extras --cpu-queries 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['value'], d['stage_ms_per_step'])" || exit 1
done
