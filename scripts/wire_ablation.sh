# the sweep on a many-facet wire with parts switched off (developer probe; needs `make -C nanokappa_amd/csrc ablate`)
# usage: wire_ablation.sh N_SIDES PARTICLES
for m in 0 8 16 24 32 64 2; do
  echo "NK_DEBUG=$m"
  NK_LIBNAME=libnanokappa_hip_ablate.so NK_DEBUG=$m timeout -k 10 200 python scripts/wire_probe.py $1 $2 2>&1 | grep -E "steps  20|rror"
done
