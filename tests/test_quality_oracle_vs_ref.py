"""Frame quality statistics (SSIM / PSNR): the restatements xo_ssim_frame / xo_plane_ssd against the numbers the REFERENCE
ENCODER itself reports per output picture (x265_picture.frameData.ssim / psnrY / psnrU / psnrV, Encoder::finishFrameStats, encoder.cpp:3160-3260)
for the pictures it reconstructed -- oracle/_ref/x265enc_* with X265ENC_DUMP (ref_encode.cpp), single-threaded so that the CTU rows are summed in
row order.  SSIM is float arithmetic; the restatement keeps the reference's order of operations, so the comparison is EXACT (tolerance 0)."""
import ctypes as C
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from depths import DEPTHS, GOLDEN_DEPTHS

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from oracle_py import Oracle  # noqa: E402


def encoder_dump(tmp_path, depth, w, h, frames, preset, extra):
    enc = os.path.join(ROOT, "oracle", "_ref", "x265enc_%d" % depth)
    if not os.path.exists(enc):
        pytest.skip("oracle/_ref/x265enc_%d not built (needs /root/reference at build time)" % depth)
    dump = str(tmp_path / "dump.bin")
    env = dict(os.environ, X265ENC_DUMP=dump, MALLOC_PERTURB_="85")
    r = subprocess.run([enc, "c", "-", str(w), str(h), str(frames), preset, str(tmp_path / "o.hevc"), "ssim=1", "psnr=1", "log-level=2"] + extra, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = open(dump, "rb").read()
    pics, off = [], 0
    ysz, csz = w * h, (w // 2) * (h // 2)
    while off < len(d):
        poc = int(np.frombuffer(d, np.int32, 1, off)[0]); off += 4
        st = np.frombuffer(d, np.float64, 4, off).copy(); off += 32
        planes = []
        for n, shape in ((ysz, (h, w)), (csz, (h // 2, w // 2)), (csz, (h // 2, w // 2))) * 2:
            planes.append(np.frombuffer(d, np.uint16, n, off).reshape(shape)); off += 2 * n
        pics.append(dict(poc=poc, ssim=st[0], psnr=st[1:4], src=planes[:3], rec=planes[3:]))
    assert len(pics) == frames
    return pics


def ssim_oracle(ora, rec, src, ctu):
    H, W = rec.shape
    t = np.uint8 if ora.depth == 8 else np.uint16
    a, b = np.ascontiguousarray(rec.astype(t)), np.ascontiguousarray(src.astype(t))
    nrows = (H + ctu - 1) // ctu
    rs, rc = np.zeros(nrows, np.float32), np.zeros(nrows, np.uint32)
    tot, cnt = C.c_double(0), C.c_uint32(0)
    P = lambda x: C.c_void_p(x.ctypes.data)
    ora.lib.xo_ssim_frame(P(a), C.c_ssize_t(W), P(b), C.c_ssize_t(W), W, H, ctu, P(rs), P(rc), C.byref(tot), C.byref(cnt))
    return rs, rc, tot.value, cnt.value


@pytest.mark.parametrize("depth,w,h,frames,preset,extra,ctu", [
    (8, 128, 64, 2, "ultrafast", [], 32),
    (8, 136, 72, 3, "medium", ["bframes=1"], 64),           # picture no CTU multiple: two CTU rows, the second 8 rows high
    (8, 200, 152, 2, "faster", ["ctu=32"], 32),             # five CTU rows, width giving a partial last group of windows
    (10, 136, 104, 2, "fast", ["ctu=16", "qp=40"], 16),     # float path of ssim_end_1
    (10, 64, 64, 2, "medium", [], 64),
])
def test_ssim_and_psnr_match_the_encoders_frame_statistics(tmp_path, depth, w, h, frames, preset, extra, ctu):
    ora = Oracle(depth)
    ora.lib.xo_plane_ssd.restype = C.c_uint64
    for pic in encoder_dump(tmp_path, depth, w, h, frames, preset, extra):
        rs, rc, tot, cnt = ssim_oracle(ora, pic["rec"][0], pic["src"][0], ctu)
        assert cnt > 0 and tot / cnt == pic["ssim"], (pic["poc"], tot / cnt, pic["ssim"])
        maxv = 255 << (depth - 8)
        for c in range(3):
            t = np.uint8 if depth == 8 else np.uint16
            a, b = np.ascontiguousarray(pic["src"][c].astype(t)), np.ascontiguousarray(pic["rec"][c].astype(t))
            ph, pw = a.shape
            ssd = int(ora.lib.xo_plane_ssd(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_ssize_t(pw), pw, ph))
            ref = float(maxv) * maxv * (w * h) / (1.0 if c == 0 else 4.0)
            exp = 10.0 * math.log10(ref / float(ssd)) if ssd else 99.99
            assert exp == pic["psnr"][c], (pic["poc"], c, exp, pic["psnr"][c])


def golden_pictures(depth):
    g = np.load(os.path.join(HERE, "golden", "quality_%d.npz" % depth))
    n = len(g["ssim"])
    return int(g["ctu"]), [dict(ssim=float(g["ssim"][i]), psnr=g["psnr"][i], src=[g["src%d_%d" % (i, c)] for c in range(3)],
                                rec=[g["rec%d_%d" % (i, c)] for c in range(3)]) for i in range(n)]


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_ssim_oracle_matches_golden_encoder_statistics(depth):
    """the same check on the committed fixtures (tests/make_golden_quality.py): runs where /root/reference is absent"""
    ora = Oracle(depth)
    ctu, pics = golden_pictures(depth)
    for pic in pics:
        rs, rc, tot, cnt = ssim_oracle(ora, pic["rec"][0], pic["src"][0], ctu)
        assert tot / cnt == pic["ssim"]
