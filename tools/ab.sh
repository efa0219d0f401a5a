# A/B on one box: bash tools/ab.sh repeats "ENV_A" "ENV_B" ...  (interleaved bench runs; prints images/s)
export TMPDIR=/tmp; mkdir -p gpurun_out
N=$1; shift
for i in $(seq $N); do
  for v in "$@"; do echo "$v: $(env $v timeout 300 python bench.py --no-cpu-baseline --no-nms --no-roofline --no-extras 2>&1 | tail -1 | cut -c84-92)"; done
done | tee gpurun_out/ab.log
