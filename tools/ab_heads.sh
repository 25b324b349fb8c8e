cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06f
for rep in 1 2 3; do
for L in libowwhip.so libowwhip_apipe.so; do
  OWW_LIB=$PWD/openwakeword_amd/$L python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']; print('$rep $L big step %.4f heads %.4f' % (d['ms_per_step'], k['heads']))" | tee -a gpurun_out/r06f/apipe_ab.txt
  OWW_LIB=$PWD/openwakeword_amd/$L python bench.py --streams 4096 --heads hey_jarvis --steps 1000 --warmup 300 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']; print('$rep $L c1 step %.4f heads %.4f' % (d['ms_per_step'], k['heads']))" | tee -a gpurun_out/r06f/apipe_ab.txt
  OWW_LIB=$PWD/openwakeword_amd/$L python bench.py --heads alexa,hey_mycroft,hey_jarvis,hey_rhasspy,timer,weather --steps 30 --warmup 10 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']; print('$rep $L six step %.4f heads %.4f' % (d['ms_per_step'], k['heads']))" | tee -a gpurun_out/r06f/apipe_ab.txt
done
done
