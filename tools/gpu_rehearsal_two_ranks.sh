timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "two_lambda_windows or both_forms or merged_localities_with_the_plain" 2>&1 | tail -3 || exit 1
BENCH_REHEARSAL_GLOO=1 timeout -k 10 600 python bench.py --gpus 2 --atoms 24k --elec rf --no-dd-leg --no-cpu-baseline --steps 20 --warmup 5 2>gpurun_out/r4m_rehearsal.err | grep '^{' > gpurun_out/r4m_rehearsal.json || { tail -20 gpurun_out/r4m_rehearsal.err; exit 1; }
python -c "
import json; r=json.load(open('gpurun_out/r4m_rehearsal.json')); print('n_gpus', r['n_gpus'], 'ms', r['ms_per_step'], 'cold', r.get('ms_per_step_cold')); print(r['lambda_window_set_over_the_ranks'])"
