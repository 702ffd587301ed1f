cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
L=soft-intro-vae-pytorch_amd/sivae_hip/libsivae_hip.so
cp $L /tmp/new.so
for round in 1 2; do
for which in new old; do
if [ $which = old ]; then cp tools/_build/libsivae_old.so $L; else cp /tmp/new.so $L; fi
for cfg in celeb128 celeb256; do
echo "== $which $cfg"
python bench.py --config $cfg --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-150
done; done; done
cp /tmp/new.so $L
