import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f)); print(f, round(d["ms_per_step"],4), "%.3e"%d["value"], {k:round(v*1e3,1) for k,v in d["kernel_ms"].items()})
    except Exception as e: print(f, "ERR", e)
