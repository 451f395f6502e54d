for kv in "" "HSA_ENABLE_INTERRUPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "HIP_FORCE_DEV_KERNARG=1" "HSA_ENABLE_INTERRUPT=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 HIP_FORCE_DEV_KERNARG=1"; do
  echo "[$kv]"; env $kv python bench.py --gpus 1 --steps 20 --warmup 5 --no-also --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(' value %.4g  ms_per_step %.5f  launch_us %.3f  fixed %.2f  wall_samples %s' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['graph_replay_fixed_cost_us'], d['config']['wall_us_per_sample']))"
done
