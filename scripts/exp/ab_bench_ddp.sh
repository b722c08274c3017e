# usage: scripts/exp/ab_bench_ddp.sh "ENV_A" "ENV_B" [bench args] -- like ab_bench.sh, through the data-parallel wrapper on ONE rank (RCCL buckets forced)
A="$1"; B="$2"; shift 2
for i in 1 2 3; do
  for v in "$A" "$B"; do
    r=$(env $v VC_DDP_FORCE_COMM=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 295$((30 + i)) bench.py --gpus 1 --no-cpu-baseline --steps 60 --warmup 15 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])") || exit 1
    echo "[$v] $r"
  done
done
