timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "degenerate or first_round or overflow or repair or fallback or topk" 2>&1 | tail -8
for S in 256; do
timeout 900 python bench.py --workload config4 --steps $S --warmup 70 --regions 2 --no-cpu-baseline 2>gpurun_out/c4_$S.err | tail -1 > gpurun_out/c4_line_$S.json
python -c "
import json; d=json.load(open('gpurun_out/c4_line_$S.json'))
print('steps', $S, 'eval bf16 ms', round(d['eval_ms_per_pass'],1), 'f32 ms', round(1e5/d['roofline_eval']['eval_users_per_s']*1e3,1), d['eval_info'], d['eval_fast_stats'], d['last_losses'][:2])"
done
