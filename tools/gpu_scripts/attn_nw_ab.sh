for nw in auto 4 8; do
  if [ "$nw" = auto ]; then unset SDM_ATTN_NW; else export SDM_ATTN_NW=$nw; fi
  timeout 200 python bench.py --timed-only --steps 2 --warmup 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('NW=$nw', d['ms_per_step'], d['kernel_breakdown_ms']['attn_d64'])"
done
