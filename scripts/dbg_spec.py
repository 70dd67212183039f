import numpy as np, torch, sys
sys.path.insert(0,'.')
from tests import oracle_lib as O
from vqengine_amd import abi, capi, synth
ctx=capi.Context(0)
eq=synth.equirect(128,64)
co,n=O.mip_chain(eq); cg,_=ctx.mip_chain(torch.from_numpy(eq).cuda())
for order in (0,1):
    sg,_=ctx.conv_specular(cg,128,64,n,32,order,abi.FMT_RGBA32F); so,_=O.conv_specular(co,128,64,n,32,order,abi.FMT_RGBA32F)
    sg=sg.cpu().numpy()
    ne=(sg.view(np.uint32)!=so.view(np.uint32))
    idx=np.argwhere(ne)
    print('order',order,'mismatch',ne.sum(),'of',ne.size)
    for i in idx[:10]:
        print(i, sg[tuple(i)], so[tuple(i)], sg[i[0]], so[i[0]])
