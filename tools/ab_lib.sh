mkdir -p gpurun_out/r3m; export ZSG_TUNE_CACHE=$PWD/gpurun_out/r3m/tune.json
B="python bench.py --no-cpu-baseline --no-bx --no-roofline --steps 100 --warmup 20"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])'
$B > /dev/null 2>&1
for i in 1 2 3 4; do
  ZSG_LIB_PATH=$PWD/zsgnet-pytorch_amd/build/abl/libzsg_oldfin.so $B 2>/dev/null | grep "^{" | python -c "$P" old
  $B 2>/dev/null | grep "^{" | python -c "$P" new
done > gpurun_out/r3m/ab.txt
cat gpurun_out/r3m/ab.txt
