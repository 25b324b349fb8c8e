# A/B of kernel build variants: libowwhip.so (default) and openwakeword_amd/libowwhip_<tag>.so
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for L in openwakeword_amd/libowwhip.so openwakeword_amd/libowwhip_*.so; do
  OWW_LIB=$PWD/$L python bench.py --steps 20 --warmup 5 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline_all']['cnn_all_stages']['frac'])"
done
OWW_PROF_BLOCK=20000 python tools/phase_profile.py
