for so in waldo_amd/lib/abl/*.so; do echo "== $so"; WALDO_HIP_LIB=$PWD/$so python tools_dev/dbg_bwd.py 2>&1 | grep -E "^8 32 48|^6 32" ; done
