run() { python bench.py --no-cpu-baseline --no-op-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms'%d['ms_per_step'], sys.argv[1:])" "$@"; }
run --set "pq_transformer._HEADS_SIDE='never'"
run --set "pq_transformer._HEADS_SIDE='never'" --set "pq_transformer._OVERLAP_KEY_SIDE='never'"
run --set "pq_transformer._HEADS_SIDE='never'" --set "decoder_rows._JOIN_PER_LAYER=False"
run --set "pq_transformer._HEADS_SIDE='never'" --set "pq_transformer._WGRAD_SIDE=False"
run --set "pq_transformer._HEADS_SIDE='never'" --set "pq_transformer._FLUSH_STREAM='sampling'"
