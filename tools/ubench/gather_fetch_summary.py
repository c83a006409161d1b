"""FETCH_SIZE (rocprofv3 --pmc) against the known byte counts of tools/ubench/gather_fetch -> the factor to apply to FETCH_SIZE
per access pattern.   python tools/ubench/gather_fetch_summary.py <dir with the counter csv> <stdout of gather_fetch>"""
import csv, glob, json, sys
d, log = sys.argv[1], sys.argv[2]
known = {}
for r in csv.DictReader(open(log)):
    known[r["kernel"]] = (int(r["useful_bytes"]), int(r["pieces"]), int(r["piece_bytes"]))
out = {}
for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k in known and r["Counter_Name"] in ("FETCH_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_REQ_sum", "TCP_TCC_READ_REQ_sum"):
            out.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
res = {}
for k, (useful, pieces, pb) in known.items():
    c = out.get(k, {})
    if "FETCH_SIZE" not in c:
        continue
    fetched = c["FETCH_SIZE"] * 1024.0
    res[k] = dict(useful_bytes=useful, piece_bytes=pb, fetch_size_bytes=int(fetched), fetch_over_useful=round(fetched / useful, 3),
                  fetch_bytes_per_piece=round(fetched / pieces, 2),
                  lines64_touched_bytes=pieces * 64 if pb <= 64 else useful,
                  fetch_over_64B_lines=round(fetched / (pieces * 64 if pb <= 64 else useful), 3), **{n: v for n, v in c.items() if n != "FETCH_SIZE"})
print(json.dumps(res, indent=1))
