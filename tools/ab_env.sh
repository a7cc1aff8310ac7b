cd /root/repo
for i in 1 2; do
for v in "X=1" "SGX_TRK_FASTX=0"; do
env $v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --many-channels 0 --concurrent 0 --no-config4 --no-from-file 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-18s track_kernel_ms %.3f  x_realtime %.1f' % ('$v', d['track_kernel_ms'], d['x_realtime']))"
done; done
