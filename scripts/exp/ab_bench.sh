# usage: scripts/exp/ab_bench.sh "ENV_A" "ENV_B" [bench args]   -- three interleaved pairs on one box
A="$1"; B="$2"; shift 2
for i in 1 2 3; do
  for v in "$A" "$B"; do
    r=$(env $v timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 60 --warmup 15 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])") || exit 1
    echo "[$v] $r"
  done
done
