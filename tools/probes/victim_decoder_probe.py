"""Run the decoder eagerly with and without a split-operand 1x1 convolution on another stream; report the first op whose output differs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from xmem2_amd import XMem, ops
from xmem2_amd.ops import ConvWeights
from xmem2_amd.synth import synthetic_state_dict
dev = torch.device('cuda:0')
torch.manual_seed(0)
os.environ['XMEM_HIP_GRAPHS'] = '0'
cfg = bench.b32_config()
net = XMem(dict(cfg), None).to(dev).eval(); net.load_weights(synthetic_state_dict(0))
side = torch.cuda.Stream()
x64 = torch.randn(4, 120, 216, 64, device=dev)
cwh = ConvWeights((torch.randn(256, 1, 1, 64) * 0.1).to(dev), torch.ones(256, device=dev), torch.zeros(256, device=dev), 1, 0)
out_hog = torch.empty(4, 120, 216, 256, device=dev)
HOG = os.environ.get('HOG', 'fp32x')
CONV = ops.conv2d
def hog(n=10):
    with torch.cuda.stream(side), ops.ws_scope('@hog'), ops.precision(HOG):
        for _ in range(n):
            CONV(x64, cwh, out=out_hog)
hog(1); torch.cuda.synchronize()
img = torch.randn(1, 480, 864, 4, device=dev); img[..., 3] = 0
key, shr, sel, f16, f8, f4 = net._encode_key_eager(img, True, True)
cat16 = torch.randn(1, 30, 54, 1600, device=dev) * 0.3
hidden0 = torch.randn(1, 30, 54, 64, device=dev) * 0.3
names = ['conv2d', 'upsample2x_add', 'cbam_residual', 'area_downsample', 'copy_channels', 'gru_gate']
orig = {n: getattr(ops, n) for n in names}
log = []
def wrap(n):
    def f(*a, **k):
        r = orig[n](*a, **k)
        t = r if isinstance(r, torch.Tensor) else None
        if t is not None:
            log.append((n, tuple(t.shape), t.clone()))
        return r
    return f
for n in names:
    setattr(ops, n, wrap(n))
def run(with_hog):
    log.clear()
    h = hidden0.clone(); c = cat16.clone()
    if with_hog:
        hog(int(os.environ.get('NHOG', '400')))
    nh, logits = net._segment_eager(f16, f8, f4, c, h, True, None, None)
    torch.cuda.synchronize()
    return list(log), logits.clone()
base, lg0 = run(False)
base2, lg1 = run(False)
print('solo vs solo: logits equal', torch.equal(lg0, lg1))
for rep in range(8):
    got, lg = run(True)
    first = None
    for i, ((n0, s0, t0), (n1, s1, t1)) in enumerate(zip(base, got)):
        if n0 != 'area_downsample' and not torch.equal(t0, t1):
            first = (i, n0, s0, float((t0 - t1).abs().max()), float((t0 != t1).float().mean()))
            break
    print(f'rep {rep}: logits equal {torch.equal(lg0, lg)}; first differing op: {first}')
    if first is not None and first[1] == 'upsample2x_add' and rep < 3:
        i = first[0]
        t0, t1 = base[i][3 - 1], got[i][3 - 1]
        skip = got[i - 1][2]
        gsrc = [x for x in got[:i] if x[1] == (1, 30, 54, 512)][-1][2]
        import torch.nn.functional as F
        bil = F.interpolate(gsrc.permute(0, 3, 1, 2), scale_factor=2, mode='bilinear', align_corners=False).permute(0, 2, 3, 1)
        bad = torch.nonzero(t0 != t1)
        print('   bad elements:', bad.shape[0], 'pixels (y,x) range', bad[:, 1].min().item(), bad[:, 1].max().item(), bad[:, 2].min().item(), bad[:, 2].max().item(),
              'channels', bad[:, 3].min().item(), bad[:, 3].max().item())
        for b in bad[:6].tolist():
            _, y, x, c = b
            print(f'   (y {y}, x {x}, c {c}): want {float(t0[0, y, x, c]):.5f} got {float(t1[0, y, x, c]):.5f}  skip {float(skip[0, y, x, c]):.5f} bilinear {float(bil[0, y, x, c]):.5f}'
                  f'  got-skip {float(t1[0, y, x, c] - skip[0, y, x, c]):.5f}  got-bil {float(t1[0, y, x, c] - bil[0, y, x, c]):.5f}')
        ys = bad[:, 1] * 108 + bad[:, 2]
        print('   distinct pixels:', torch.unique(ys).numel(), ' first flat pixel ids:', torch.unique(ys)[:10].tolist())
