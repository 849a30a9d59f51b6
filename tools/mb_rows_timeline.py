"""Cycle-counter timeline of spconv_fwd_rows<64,64> (library built with -DSPR_TIMELINE=1):
    make -C vision3d_amd/csrc clean && make -C vision3d_amd/csrc EXTRA=-DSPR_TIMELINE=1
stamps per (workgroup 5/133/261/389, wave): 0 start, 1 nbr staged, 2/3 first operand loads issued, 4..10 after each
multiply, 12 partials stored, 13 after the barrier, 14 end."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision3d_amd import _lib as L

raw = ctypes.CDLL(L.LIB_PATH)
if not hasattr(raw, "v3d_debug_rows_timeline"):
    sys.exit("library built without -DSPR_TIMELINE=1")
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8160
K, C = 27, 64
x = torch.randn(n, C, device="cuda")
w = torch.randn(K, C, C, device="cuda") / 40
nbr = torch.randint(0, n, (K, n), device="cuda", dtype=torch.int32)
nbr[torch.rand(K, n, device="cuda") > 0.31] = -1
img = torch.empty(int(L.lib().v3d_sparse_conv_weight_image_bytes(K, C, C)), dtype=torch.uint8, device="cuda")
L.check(L.lib().v3d_sparse_conv_pack_weights(L.ptr(w), K, C, C, 0, L.ptr(img), L.stream_ptr()), "pack")
n_dev = torch.tensor([n], dtype=torch.int32, device="cuda")
out = torch.empty(n, C, device="cuda")
for variant in (10,):
    for _ in range(5):
        L.check(L.lib().v3d_sparse_conv_fwd_packed(L.ptr(x), L.ptr(img), L.ptr(nbr), L.ptr(n_dev), n, K, C, C, None, None, 0,
                                                   L.ptr(out), -variant, 0, None, None, None, None, None, L.stream_ptr()), "fwd")  # negative rows_hint = forced kernel
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 256)()
    raw.v3d_debug_rows_timeline(buf)
    t = np.array(list(buf), dtype=np.int64).reshape(4, 4, 16)
    print("variant", variant)
    for b in range(4):
        base = t[b, :, 0].min()
        for wv in range(4):
            print(f"  wg{5 + 128 * b:4d} wave{wv}", [int(v - base) if v else -1 for v in t[b, wv, :15]])
