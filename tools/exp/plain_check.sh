cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
timeout 600 python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 8 --warmup 3 --no_cpu_baseline --legs joint 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('plain  regression %7.2f img/s %8.3f ms | joint %7.2f img/s %8.3f ms | host enqueue %s, gpu still queued %s' % (j['value'], j['ms_per_step'], j['joint']['value'], j['joint']['ms_per_step'], j['joint'].get('host_enqueue_ms_per_step'), j['joint'].get('gpu_still_queued_when_host_is_done_ms')))"
done
