import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, collections
import procedural as P
from oaprogressionmmf_amd import ops
from oaprogressionmmf_amd.models._core_fes import dict_fes
from oaprogressionmmf_amd.models._encoder import KoafTrunk
import oaprogressionmmf_amd.models._encoder as E
dev=torch.device('cuda:0')
cnt=collections.Counter()
orig_ap=ops.act_planes
def ap(x,npix,C,tf=0,*a,**k):
    cnt[f'act_planes tf{tf}']+=1
    return orig_ap(x,npix,C,tf,*a,**k)
ops.act_planes=ap
orig_cf=ops.conv2d_fwd
def cf(*a,**k):
    if k.get('emit') is not None: cnt['emit']+=1
    return orig_cf(*a,**k)
ops.conv2d_fwd=cf
net=dict_fes['resnet50'](pretrained=False)
trunk=KoafTrunk(*list(net.children())[:-1]); P.fill_state_dict(trunk.state_dict()); trunk=trunk.to(dev).train()
from oaprogressionmmf_amd.arena import get_arena
get_arena(trunk)
for rec in (False,[0,1,2]):
    trunk.recompute=rec
    cnt.clear()
    x=torch.randn(4,1,128,128,device=dev)
    y=trunk(x); print('fwd',dict(cnt)); cnt.clear()
    y.sum().backward(); torch.cuda.synchronize(); print('rec',rec,'bwd',dict(cnt))
print('takes', E._takes_planes(net.layer1[0].conv2), E.EMIT_PLANES, ops.APLANES_MASK)
