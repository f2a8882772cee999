for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-cold-start --no-solve --sqp-iterations 0 --reps 3 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('run $i', r.get('host_api_ms_per_sweep'), r.get('host_api_path'), r.get('ms_per_step'))"
done
OGPSX_TIMING=1 python bench.py --no-cpu-baseline --no-cold-start --no-solve --sqp-iterations 0 --reps 3 2>&1 | grep "ogpsx timing" | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold-start --no-solve --sqp-iterations 0 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('driver style', r.get('host_api_ms_per_sweep'), r.get('host_api_path'), r.get('ms_per_step'))"
OGPSX_TIMING=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold-start --no-solve --sqp-iterations 0 2>&1 | grep "ogpsx timing" | tail -3
