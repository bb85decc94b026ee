import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import bench, libsrcnn_amd as S
TICK = os.sysconf("SC_CLK_TCK")
def snap():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try: f = open("/proc/self/task/%s/stat" % t).read()
        except OSError: continue
        rest = f[f.rindex(")") + 2:].split()
        out[int(t)] = (int(rest[11]), int(rest[12]))
    return out
S.init(0)
step, free = bench.host_stream_setup(S, 16)
step()
a = snap(); c0 = time.process_time(); t0 = time.perf_counter()
for _ in range(4): step()
wall = time.perf_counter() - t0; cpu = time.process_time() - c0; b = snap()
rows = sorted(((b[t][0] - a.get(t, (0, 0))[0], b[t][1] - a.get(t, (0, 0))[1], t) for t in b), key=lambda r: -(r[0] + r[1]))[:3]
print("wall %.2f ms/frame, cpu %.1f ms/frame; top threads (user, sys ms/frame): %s" % (wall * 1e3 / 64, cpu * 1e3 / 64, [(round(u / TICK * 1e3 / 64, 1), round(s / TICK * 1e3 / 64, 1)) for u, s, t in rows]))
# which syscall is the hottest long-lived thread in?  sample it while another batch runs
import threading
hot = rows[0][2]; seen = {}
stop = False
def sampler():
    while not stop:
        try:
            sc = open("/proc/self/task/%d/syscall" % hot).read().split()[0]
            wc = open("/proc/self/task/%d/wchan" % hot).read().strip()
            st = open("/proc/self/task/%d/stat" % hot).read(); st = st[st.rindex(")") + 2]
        except OSError: sc, wc, st = "?", "?", "?"
        seen[(sc, wc, st)] = seen.get((sc, wc, st), 0) + 1
        time.sleep(0.0005)
th = threading.Thread(target=sampler); th.start()
for _ in range(2): step()
stop = True; th.join()
print("hot thread %d samples (syscall nr, wchan, state): %s" % (hot, sorted(seen.items(), key=lambda kv: -kv[1])[:6]))
free()
