set -e
for u in 2 3 4 6; do
  echo "== Q1 group U=$u"; VDL_JIT=1 VDL_JIT_GROUP_U=$u python bench.py --query q1 --steps 10 --warmup 3 --no-cpu-baseline --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], d['roofline']['kernel_us'], d['roofline']['frac'])"
done
for u in 4 6 8 12; do
  echo "== Q6 via mscan U=$u"; VDL_JIT=1 VDL_NO_KSCAN=1 VDL_JIT_U=$u python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], d['roofline']['kernel_us'], d['roofline']['frac'], d['verified_bit_exact_vs_cpu'])"
done
echo "== Q6 k_scan"; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], d['roofline']['kernel_us'], d['roofline']['frac'])"
