# A/B of kernel build variants: libowwhip.so (default) and openwakeword_amd/libowwhip_<tag>.so
for L in openwakeword_amd/libowwhip.so openwakeword_amd/libowwhip_*.so; do
  OWW_LIB=$PWD/$L python bench.py --steps 12 --warmup 4 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$L', d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline_all']['cnn_all_stages']['frac'])"
done
