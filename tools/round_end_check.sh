#!/bin/bash
# everything the driver runs at round end, in one go: GPU test suite, smoke, the default bench line
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_final; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -n 4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 2 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_final/bench.json').read().strip().splitlines()[-1])
rl=d['roofline']
print('value %.4g steps %d ms/step %.4f' % (d['value'], d['steps'], d['ms_per_step']))
print('roofline', {k: rl.get(k) for k in ('bound','contract_bound','achieved','frac','traffic','measured_hbm_frac','lds_frac','iterate_frac')})
print('round', {k: rl.get('round_split', {}).get(k) for k in ('iterate_and_store','poll','apply_and_barrier','round_us_from_solve')})
print('other', {k: (v.get('us_per_iteration'), v.get('iterate_frac')) for k, v in d.get('other_configs', {}).items() if isinstance(v, dict)})
print('facade', {k: v.get('update_ms_p50') for k, v in d['facade_frame_ms'].items() if isinstance(v, dict)})
print('frames_axis', d['frames_axis']['frame_iterations_per_s'], 'host_inclusive', d['host_inclusive']['ms_per_frame'])
print('small', d.get('small_graph_us_per_iteration'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('cores'), 'parity', d['parity_vs_oracle']['bit_exact'])
PY
