"""How round 5 located the back end's read of unwritten device memory (output of the original run: profiles/r05_stale_memory.txt).

The tree at hand no longer failed, so the failure was dug out where it last showed: commit cc98ac6 introduced the create-time memsets that cured it.

  python scripts/stale_archaeology.py prepare     (CPU, here: git archive cc98ac6 -> tmp_arch/Y, the memsets behind switches, build for gfx950 -- ~1 min)
  python scripts/stale_archaeology.py ddmin       (GPU box: reproduce, then delta-debug which allocations must stay un-zeroed -- ~4 min)

Switches patched into tmp_arch/Y/ground-fusion_amd/csrc/gf_ba.hip (Buf::alloc) and gf_tracker.hip:
  GF_NOZERO=1                 no create-time memset at all (the tree as it was before cc98ac6)
  GF_ZERO_FROM_HANDLE=k       the set below applies to gf_ba handles number >= k only (earlier handles stay un-zeroed: they are what leaves the stale data behind)
  GF_NOZERO_SET=a,b,c|none    per-handle allocation ordinals that stay un-zeroed (everything else of those handles is zeroed)
  GF_BA_HANDLE_TRACE=1 / GF_BA_ALLOC_TRACE=1   handle numbers / the allocation statement behind every ordinal
Reproducer (18 s): tests/test_estimator_gpu.py -k "test_replay_with_gnss_matches_oracle or test_group_with_gnss_members" in ONE process fails in
test_group_with_gnss_members (a member ran 8 iterations where the stand-alone estimator ran 3) with GF_NOZERO=1 and passes without.
Result of the delta debugging: the minimal set is {29} = h->pri_c.alloc(B, false)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Y = os.path.join(ROOT, "tmp_arch", "Y")
COMMIT = "cc98ac6"
K = "test_replay_with_gnss_matches_oracle or test_group_with_gnss_members"


def prepare():
    os.makedirs(Y, exist_ok=True)
    subprocess.check_call("git archive %s | tar -x -C %s" % (COMMIT, Y), shell=True, cwd=ROOT)
    p = os.path.join(Y, "ground-fusion_amd", "csrc", "gf_ba.hip")
    s = open(p).read()
    old = 'if (hipMemset(d, bad ? 0x5A : 0, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return gf::set_err(GF_ERR_HIP, "hipMemset failed");'
    new = '''{
            const int ord = g_handle_ord++;
            bool zero = !getenv("GF_NOZERO") || bad;
            const int from_handle = getenv("GF_ZERO_FROM_HANDLE") ? atoi(getenv("GF_ZERO_FROM_HANDLE")) : 0;
            if (ord == 0 && getenv("GF_BA_HANDLE_TRACE")) fprintf(stderr, "gf_ba handle %d created\\n", g_handle_no);
            if (const char* z = g_handle_no >= from_handle ? getenv("GF_NOZERO_SET") : nullptr) {
                zero = true;
                for (const char* q = z; *q; ) { if (*q >= '0' && *q <= '9') { if (atoi(q) == ord && !bad) zero = false; while (*q >= '0' && *q <= '9') q++; } else q++; }
            }
            if (getenv("GF_BA_ALLOC_TRACE") && g_handle_no < 2) fprintf(stderr, "gf_ba handle %d ord %d: %s  %zu x %zu B%s\\n", g_handle_no, ord, g_alloc_what ? g_alloc_what : "?", count, sizeof(T), zero ? " zeroed" : "");
            if (zero && (hipMemset(d, bad ? 0x5A : 0, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess)) return gf::set_err(GF_ERR_HIP, "hipMemset failed");
        }'''
    assert old in s
    s = s.replace(old, new)
    s = s.replace("namespace {\nstd::atomic<long long> g_up_bytes{0}, g_up_calls{0};", "namespace {\nint g_handle_ord = 0, g_handle_no = -1; const char* g_alloc_what = nullptr;\nstd::atomic<long long> g_up_bytes{0}, g_up_calls{0};")
    s = s.replace("#define A_(x) do { if (int rc_ = (x))", "#define A_(x) do { g_alloc_what = #x; if (int rc_ = (x))")
    s = s.replace("    gf_ba* h = new gf_ba();\n    h->cfg = *cfg;", "    gf_ba* h = new gf_ba();\n    g_handle_ord = 0; g_handle_no++;\n    h->cfg = *cfg;")
    open(p, "w").write(s)
    p = os.path.join(Y, "ground-fusion_amd", "csrc", "gf_tracker.hip")
    s = open(p).read()
    old = "if (hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess"
    assert old in s
    open(p, "w").write(s.replace(old, 'if (!getenv("GF_NOZERO") && hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess'))
    subprocess.check_call([sys.executable, os.path.join("ground-fusion_amd", "build.py")], cwd=Y)
    subprocess.check_call(["make", "-C", "oracle"], cwd=Y)
    print("prepared", Y, "(tmp_arch/ is git-ignored and travels to the GPU box with gpurun)")


def probe(unz, log):
    e = dict(os.environ, GF_NOZERO_SET=",".join(str(x) for x in sorted(unz)) if unz else "none")
    cmd = [sys.executable, "-m", "pytest", "tests/test_estimator_gpu.py", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", K]
    t = time.time()
    r = subprocess.run(cmd, cwd=Y, env=e, capture_output=True, text=True)
    failed = "test_group_with_gnss_members" in r.stdout and "FAILED" in r.stdout
    log("   probe |unzeroed| = %2d %s -> %s (%.0f s)" % (len(unz), sorted(unz) if len(unz) <= 12 else "", "FAILS" if failed else "passes", time.time() - t))
    return failed


def ddmin(budget=900.0):
    t0 = time.time()

    def log(*a):
        print(" ".join(str(x) for x in a), flush=True)
    e = dict(os.environ, GF_NOZERO="1", GF_BA_HANDLE_TRACE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_estimator_gpu.py", "-m", "gpu", "-q", "-x", "-s", "-p", "no:cacheprovider", "-k", K], cwd=Y, env=e, capture_output=True, text=True)
    hs = [int(ln.split()[2]) for ln in (r.stderr + "\n" + r.stdout).splitlines() if ln.startswith("gf_ba handle ") and ln.endswith("created")]
    log("handles created:", hs, "| fails with GF_NOZERO=1:", "FAILED" in r.stdout)
    os.environ["GF_NOZERO"] = "1"
    os.environ["GF_ZERO_FROM_HANDLE"] = str(max(hs) - 3)      # the last four handles belong to test_group_with_gnss_members (one group handle + three stand-alone estimators)
    cur, n = list(range(73)), 2
    if not probe(cur, log) or probe([], log):
        log("not reproduced / not in these handles' buffers")
        return
    while len(cur) >= 2 and time.time() - t0 < budget:      # ddmin (Zeller)
        chunk = max(1, len(cur) // n)
        subsets = [cur[i:i + chunk] for i in range(0, len(cur), chunk)]
        reduced = False
        for sset in subsets:
            if probe(sset, log):
                cur, n, reduced = sset, 2, True
                break
        if not reduced:
            for sset in subsets:
                comp = [x for x in cur if x not in sset]
                if comp and probe(comp, log):
                    cur, n, reduced = comp, max(n - 1, 2), True
                    break
        if not reduced:
            if n >= len(cur):
                break
            n = min(len(cur), 2 * n)
    log("minimal failing set of un-zeroed ordinals:", cur, "in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    {"prepare": prepare, "ddmin": ddmin}[sys.argv[1]]()
