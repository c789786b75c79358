"""in-kernel stamps of the F(4x4) kernel (needs a build with -DW4_STAMPS: make -C imgcomp_cvpr_amd/csrc clean all CXXFLAGS+=-DW4_STAMPS): prologue / loop / epilogue shader clocks per wave"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'wino4_check.py')).read().split("torch.set_num_threads(16)")[0])
wgs = int(lib.ic_wino4_3x3_c128_workgroups(N, H, W))
waves = int(os.environ.get('W4_WAVES', '8'))
raw = ctypes.CDLL(L.LIB_PATH)
buf = torch.zeros((wgs * waves, 4), dtype=torch.int64, device=dev)
raw.ic_wino4_debug_set_buffer(ctypes.c_void_p(buf.data_ptr()))
for _ in range(5):
    run4(1, (r1d,))
torch.cuda.synchronize()
b = buf.cpu().double()
print('waves', b.shape[0], ' prologue %.0f  loop %.0f  epilogue %.0f clocks (means);  loop min %.0f max %.0f; ideal MFMA issue 36864 per wave' % (
    b[:, 0].mean(), b[:, 1].mean(), b[:, 2].mean(), b[:, 1].min(), b[:, 1].max()))
t0 = b[:, 3]
span = (t0.max() - t0.min() + b[:, :3].sum(1).max())
print('launch span ~%.0f clocks' % span)
