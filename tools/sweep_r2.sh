for cfg in "DG16_PREP_OVERLAP=0" "DG16_PREP_OVERLAP=1"; do
  echo "== $cfg"; env $cfg timeout 120 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms/proof %.3f  g2acc %.3f g1acc %.3f'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['g1_accumulate_ms']))"
done
