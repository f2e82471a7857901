"""Ad-hoc GPU sanity script (not a pytest file): first contact of every kernel with the reference."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))      # (_ref, _oracle)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import _ref, audiowmark_amd as awm

torch.cuda.init()
ctx = awm.Context(0)
rng = np.random.default_rng(1)
PAY = "0123456789abcdef0011223344556677"

def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

# 1. fft_range
for C in (1, 2, 3):
    x = rng.uniform(-1, 1, (40000, C)).astype(np.float32)
    g = ctx.fft_range(torch.from_numpy(x).cuda(), 123, 20).cpu().numpy()
    r = _ref.fft_range(x, C, 123, 20)
    print("fft_range C=%d rel err %.3g" % (C, rel(g, r)))

# 2. add
for C, secs in ((2, 130), (1, 20), (2, 0.5)):
    n = int(44100 * secs) + 17
    x = rng.uniform(-1, 1, (n, C)).astype(np.float32)
    for nolim in (True, False):
        awm.set_params(test_no_limiter=nolim); _ref.set_params(test_no_limiter=nolim)
        t0 = time.time(); r = _ref.add(None, x, C, PAY).reshape(n, C); t1 = time.time()
        g = ctx.add_watermark(None, PAY, torch.from_numpy(x).cuda()).cpu().numpy()
        d = g.astype(np.float64) - r
        print("add C=%d n=%d nolim=%d: rms %.3g max %.3g  (delta rms %.4g) ref %.2fs exact %.1f%%" % (
            C, n, nolim, np.sqrt((d ** 2).mean()), np.abs(d).max(), np.sqrt(((r - x) ** 2).mean()), t1 - t0, 100 * (d == 0).mean()))
awm.set_params(); _ref.set_params()

# 3. watermark 130 s stereo and search
n = 44100 * 130
x = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
w = _ref.add(None, x, 2, PAY).reshape(n, 2)
wd = torch.from_numpy(w).cuda()
db_g, have_g = ctx.sync_fft(wd, 256, 50)
db_r, have_r = _ref.sync_fft(w, 2, 256, 50)
print("sync_fft max abs err %.3g (values ~%.1f) have eq %s" % (np.abs(db_g.cpu().numpy() - db_r).max(), np.abs(db_r).mean(), (have_g.cpu().numpy() == have_r).all()))

t0 = time.time(); ri, rq, rm = _ref.search_approx(None, w, 2); t1 = time.time()
gi, gq, gm = ctx.search_approx(None, wd); torch.cuda.synchronize(); t2 = time.time()
print("search_approx: n %d/%d idx eq %s  raw err %.3g mean err %.3g  ref %.2fs gpu %.3fs" % (len(gi), len(ri), (gi == ri).all(), np.abs(gq - rq).max(), np.abs(gm - rm).max(), t1 - t0, t2 - t1))

t0 = time.time(); ri, rq, rb = _ref.sync_search(None, w, 2); t1 = time.time()
gi, gq, gb = ctx.sync_search(None, wd); t2 = time.time()
print("sync_search ref:", list(zip(ri.tolist(), np.round(rq, 6).tolist(), rb.tolist())))
print("sync_search gpu:", list(zip(gi.tolist(), np.round(gq, 6).tolist(), gb.tolist())))
print("  ref %.2fs gpu %.3fs" % (t1 - t0, t2 - t1))

# 4. soft bits
idxs = [int(i) for i in ri]
sb, ok = ctx.block_soft_bits(None, wd, idxs)
for i, ix in enumerate(idxs):
    r = _ref.mix_decode(None, w, 2, ix)
    if r is None:
        print("soft bits idx %d: ref empty, ok=%d" % (ix, ok[i]))
    else:
        print("soft bits idx %d: max abs err %.3g (|v| mean %.2f) ok=%d" % (ix, np.abs(sb[i] - r).max(), np.abs(r).mean(), ok[i]))

# 5. viterbi
for bt in (0, 1, 2):
    bits = rng.integers(0, 2, 128)
    coded = awm.conv_encode(bt, bits).astype(np.float32)
    soft = np.clip(coded + rng.normal(0, 0.45, coded.shape), -0.5, 1.5).astype(np.float32)
    t0 = time.time(); rbits, rerr = _ref.conv_decode_soft(bt, soft); t1 = time.time()
    gbits, gerr = ctx.viterbi_decode(bt, soft); t2 = time.time()
    print("viterbi bt=%d bits eq %s (orig eq %s) err ref %.9g gpu %.9g  ref %.3fs gpu %.3fs" % (bt, (gbits[0] == rbits).all(), (rbits == bits).all(), rerr, gerr[0], t1 - t0, t2 - t1))

# 6. decode chunk / get
t0 = time.time(); rp = _ref.decode_chunk(None, w, 2, True); t1 = time.time()
gp = ctx.decode_chunk(None, wd, True); t2 = time.time()
print("decode_chunk ref %.2fs gpu %.3fs" % (t1 - t0, t2 - t1))
for p in rp: print("  ref", p)
for p in gp: print("  gpu", p)

# 7. short clip
nclip = 44100 * 25
wc = w[44100 * 40: 44100 * 40 + nclip].copy()
rp = _ref.get(None, wc, 2)
gp = ctx.get_watermark(None, torch.from_numpy(wc).cuda())
for p in rp: print("  clip ref", p)
for p in gp: print("  clip gpu", p)
