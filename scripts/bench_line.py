import sys, json
for line in sys.stdin:
    if line.startswith('{"metric"'):
        d = json.loads(line)
        print(sys.argv[1] if len(sys.argv) > 1 else "", "%.0f" % d["value"], {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}, d.get("step_ms"), flush=True)
