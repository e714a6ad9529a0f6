import sys, json
rows = {}
for l in sys.stdin:
    try:
        d = json.loads(l)
    except Exception:
        print(l.strip()); continue
    rows.setdefault(d["case"], {}).update(d)
for k, d in rows.items():
    print(f"{k:30s} burst {d.get('burst_median_us', 0):8.1f} us {d.get('burst_TFLOPs', 0):7.1f} TF  single {d.get('median_us', 0):8.1f} us  bad={d.get('checked_bad')} err={d.get('max_err')}")
