for d in 0 64; do echo "== DBG $d"; DDMI_CONV_DBG=$d DDMI_CONV_TS=1 timeout 120 python tools/bench_conv.py --batch 256 --cfgs 11 --only "448->1344" 2>&1 | grep -A3 "conv timeline"; done
