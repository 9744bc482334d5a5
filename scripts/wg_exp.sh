for mp in 512 256 128; do for bm in 0 64 128; do echo "== MINPIX=$mp BM=$bm"; FI_WG_MINPIX=$mp FI_WG_BM=$bm python scripts/conv_bench.py "C" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-22s wgrad %7.1f us %6.1f TF/s'%(d['layer'], d['wgrad_us'], d['wgrad_TFLOPs']))
"; done; done
