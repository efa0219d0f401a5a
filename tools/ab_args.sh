# A/B of bench.py ARGUMENTS on one box: bash tools/ab_args.sh repeats "args A" "args B" ...
export TMPDIR=/tmp; mkdir -p gpurun_out
N=$1; shift
for i in $(seq $N); do
  for v in "$@"; do echo "[$v]: $(timeout 300 python bench.py --no-cpu-baseline --no-nms --no-roofline --no-extras $v 2>&1 | tail -1 | cut -c84-92)"; done
done | tee gpurun_out/ab_args.log
