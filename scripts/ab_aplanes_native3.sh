for m in 7 0 3 4; do echo "== KOAF_APLANES=$m"; KOAF_APLANES=$m timeout -k 10 200 python3 bench.py --workload native3 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r['achieved'], r['kernel_ms_per_step'])"; done
