R=${GRAFT_REPO_ROOT:-/root/repo}
for L in rows tiled; do
echo "== $L 1M x 2048"; LAYOUT=$L QS=1,16,32,64,100,128,256,1000 python3 $R/tools/bench_search.py 2>&1 | grep "^Q="
echo "== $L 1M x 4096"; LAYOUT=$L D=4096 QS=1,100,256 python3 $R/tools/bench_search.py 2>&1 | grep "^Q="
echo "== $L 10M x 256"; LAYOUT=$L N=10000000 D=256 QS=1,100,1000 python3 $R/tools/bench_search.py 2>&1 | grep "^Q="
done
