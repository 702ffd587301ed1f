cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2c
for cfg in celeb256; do
for pp in "1 1" "1 0" "0 1" "0 0" "1 1" "1 0"; do
set -- $pp
echo "== $cfg PERSIST=$1 STORE16=$2"
SIVAE_BF16_CONV_PERSIST=$1 SIVAE_BF16_CONV_STORE16=$2 python tools/bench_conv16.py $cfg 128 fwd 2>&1 | grep -E "total|64->  64|64-> 128 @128 k3|128-> 128|256-> 256"
done; done
