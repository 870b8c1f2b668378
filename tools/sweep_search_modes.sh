set -e
for M in 1 2; do
echo "== MODE $M  1M x 2048"; MODE=$M QS=1,16,32,48,64,100,128,256,1000 timeout 300 python tools/bench_search.py 2>&1 | grep "^Q="
echo "== MODE $M  1M x 4096"; MODE=$M D=4096 QS=1,32,64,100,256 timeout 300 python tools/bench_search.py 2>&1 | grep "^Q="
echo "== MODE $M  10M x 256"; MODE=$M N=10000000 D=256 QS=1,4,16,100,256,1000 timeout 300 python tools/bench_search.py 2>&1 | grep "^Q="
done
