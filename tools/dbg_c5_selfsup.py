"""Diagnosis (GPU box): configs[4] self-supervised step against the golden and the oracle, every quantity printed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import prifit_oracle as orc
import dgcnn_common as C
from prifit_amd.src import dgcnn as D
from prifit_amd.convex_loss import convex_loss
g = np.load(os.path.join(ROOT, "tests/golden/step_dgcnn_selfsup.npz"))
d = C.selfsup_inputs(g)
ref = C.selfsup_state(g, orc.OracleDGCNGn)
net = D.get_model(50, k=20)
net.net.load_state_dict(ref.state_dict())
net.cuda()
R = d["R"].cuda()
cid = torch.from_numpy(np.asarray(g["center_ids"])).long()
emb_o, _ = ref(d["xyz"])
emb_h, _ = net.net(d["xyz"].cuda())
print("embedding HIP vs oracle fp32: rel L2 %.3e  max abs %.3e" % (float((emb_h.cpu() - emb_o).norm() / emb_o.norm()), float((emb_h.cpu() - emb_o).abs().max())))
kw = dict(quantile=C.Q, iterations=C.ITERS, max_num_clusters=25, canonical=True)
# (1) oracle loss on the oracle embedding, (2) oracle loss on the HIP embedding, (3) HIP loss on the oracle embedding, (4) HIP on HIP
for name, e in (("oracle emb", emb_o.detach()), ("HIP emb", emb_h.detach().cpu())):
    to, co, po, lo, info = orc.convex_loss(d["xyz"], d["cham"], e.permute(0, 2, 1), rand_table=[[d["R"]] * 64] * 2, center_ids=d["center_ids"], return_info=True, **kw)
    th, ch, ph, lh, infoh = convex_loss(d["xyz"].cuda(), d["cham"].cuda(), e.permute(0, 2, 1).contiguous().cuda(), rand_table=R, center_ids=cid, return_info=True, **kw)
    print("%s: oracle loss %.9f  HIP loss %.9f  rel %.2e   K %s / %s  bw %s / %s" % (name, float(to), float(th), abs(float(to) - float(th)) / float(to),
          [len(p) for p in po], [len(p) for p in ph], [round(float(c["bw"]), 6) for c in info["cluster"]], [round(float(x), 6) for x in infoh["cluster"]["bw"].cpu()]))
    print("   parts oracle", [[round(float(x), 9) for x in p] for p in info["parts"]], " HIP", [[round(float(x), 9) for x in p.cpu()] for p in infoh["parts"]])
    for b in range(2):
        print("   labels equal shape %d: %s" % (b, bool(torch.equal(lo[b], lh[b].cpu().long()))))
print("golden loss %.9f" % float(np.asarray(g["total_loss"]).reshape(-1)[0]))
