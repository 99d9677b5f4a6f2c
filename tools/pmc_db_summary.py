"""Sum rocprofv3 --pmc counters per kernel name from a rocpd .db (ROCm 7 default output).  usage: pmc_db_summary.py results.db [name-filter]"""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
filt = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda key: [t for t in tabs if t.startswith(key)][0]
pmcinfo = {r[0]: r[1] for r in cur.execute(f"select id, name from {T('rocpd_info_pmc')}")}
kd = T('rocpd_kernel_dispatch'); ks = T('rocpd_info_kernel_symbol')
kcols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
disp = {r[0]: (names.get(r[1], str(r[1])), r[2], r[3]) for r in cur.execute(f"select event_id, kernel_id, start, end from {kd}")}
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.Counter()
seen = set()
for ev, pid, val in cur.execute(f"select event_id, pmc_id, value from {T('rocpd_pmc_event')}"):
    if ev not in disp: continue
    n = disp[ev][0]
    if filt not in n: continue
    agg[n][pmcinfo[pid]] += val
    if ev not in seen:
        seen.add(ev); cnt[n] += 1; dur[n] += disp[ev][2] - disp[ev][1]
for n in agg:
    print(n[:100], "dispatches", cnt[n], "avg_us", dur[n] / cnt[n] / 1e3)
    for k, v in sorted(agg[n].items()):
        print(f"    {k:36s} {v / cnt[n]:16.1f} per dispatch")
