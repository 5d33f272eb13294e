# bench.py's multi-rank launch path rehearsed on ONE GPU: two ranks under torchrun, gloo rendezvous, unique-id broadcast,
# barriers, max of the elapsed times -- with NK_COMM_DRYRUN, i.e. rank-dependent emission but no RCCL communicator (two
# ranks cannot share a device in RCCL), so the tallies stay local and `value` is not a measurement.
NK_COMM_DRYRUN=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 20 --warmup 5 --particles 1e6
