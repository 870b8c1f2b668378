# A/B of a kernel change on one box: build the changed sources into lightretriever_amd/build/liblrx_new.so
#   python -m lightretriever_amd.build --out=$PWD/lightretriever_amd/build/liblrx_new.so   (the product liblrx.so = the baseline)
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
echo "== product"; python3 $R/tools/bench_gemm.py 2>&1 | grep -E "^(o|down|qkv|gate)"
echo "== variant"; LRX_LIB_DEV_VARIANT=$R/lightretriever_amd/build/liblrx_new.so python3 $R/tools/bench_gemm.py 2>&1 | grep -E "^(o|down|qkv|gate)"
done
