import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); r=d["roofline"]
        print(f, d["value"], d["ms_per_step"], r["frac"], r["launches_per_step"], r["conv_ms_per_step"], d["host_enqueue_ms_per_step"])
    except Exception as e:
        print(f, "failed", e)
