cd "$GRAFT_REPO_ROOT"
B="--no-families --no-train-leg --no-cpu-baseline --no-f32mfma-leg --no-make-gifs-leg --no-extra-legs --no-roofline --steps 18 --warmup 3"
for cfg in "64 3" "64 1" "128 1" "128 2" "128 3" "32 3" "32 6"; do
  set -- $cfg
  timeout -k 10 300 python3 bench.py $B --batch $1 --inflight $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('batch $1 inflight $2: ', d['value'], 'frames/s', d['ms_per_step'], 'ms per rollout; single chain', d['single_chain']['ms_per_step'], 'ms')"
done
