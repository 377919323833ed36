for v in cold c16k c8k; do echo "== $v"; VOGE_HIP_LIB=$PWD/build/variants/$v.so python tools/band_time.py 2>&1 | grep -E "n=4|n=8"; done
