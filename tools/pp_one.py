"""One 3 x 3 convolution shape of the 12-row launch under the ping-pong kernel and its ablations (tuning igemm_vpp 0..4: each is its own
template instance, so a rocprofv3 kernel trace / --pmc pass separates them by name).  usage: pp_one.py [cfg] [n]"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Ctx, ptr  # noqa: E402
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 17
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = Ctx(); lib = ctx.lib
B, cin, hw, cout = 12, 320, 64, 320
x = torch.randn(B, hw, hw, cin, device="cuda").half(); w = (torch.randn(cout, 9 * cin, device="cuda") / math.sqrt(9 * cin)).half()
bias = torch.randn(cout, device="cuda"); out = torch.empty(B, hw, hw, cout, device="cuda", dtype=torch.half)
for vpp in [int(v) for v in os.environ.get("VPP", "0,1,2,3,4").split(",")]:
    assert lib.pnpi_set_tuning(b"igemm_vpp", vpp) == 0
    for _ in range(n):
        ctx.call("pnpi_op_conv", ptr(x), None, cin, 0, B, hw, hw, 3, 1, 1, 0, hw, hw, ptr(w), ptr(bias), None, cout, ptr(out), cfg, 0)
    torch.cuda.synchronize()
lib.pnpi_set_tuning(b"igemm_vpp", 0)
