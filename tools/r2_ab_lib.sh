# A/B of the working-tree library against tools/_build/libsivae_old.so (built from another commit) inside ONE gpurun call
# usage: bash tools/r2_ab_lib.sh "<bench.py args>" ["<second set of args>" ...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
L=soft-intro-vae-pytorch_amd/sivae_hip/libsivae_hip.so
cp $L /tmp/new.so
for round in 1 2; do
for which in new old; do
if [ $which = old ]; then cp tools/_build/libsivae_old.so $L; else cp /tmp/new.so $L; fi
for a in "$@"; do
echo "== $which $a"
python bench.py $a --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-150
done; done; done
cp /tmp/new.so $L
