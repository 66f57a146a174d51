import sys, os, time, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/transductive-clip_amd'); sys.path.insert(0,'/root/repo/transductive-clip_amd/drop_in')
from tclip_amd import engine
dev=torch.device('cuda:0')
for name in sys.argv[1:]:
    g=np.load(f'/root/repo/tests/golden/{name}.npz'); kind=str(g['kind']); few=kind.startswith('fs'); K=int(g['K'])
    t=time.time()
    res=engine.run_em_dirichlet(torch.from_numpy(g['x_q']).to(dev), torch.from_numpy(g['x_s']).to(dev) if few else None, torch.from_numpy(g['y_s']).to(dev) if few else None,
        n_batches=1, iters=int(g['iters']), iter_mm=1000, lambd=int(K/5)*75, hard=kind.endswith('hard'))
    torch.cuda.synchronize(); dt=time.time()-t
    mm=res.mm_iters.cpu().numpy()[0]
    out=[name, f'{dt:.2f}s', 'mm_eq', np.array_equal(mm,g['mm_iters']), 'argmax_eq', np.array_equal(res.preds.cpu().numpy(), g['argmax'][-1])]
    if 'alpha' in g.files:
        a=g['alpha']; d=res.alpha.cpu().numpy()-a
        fro=(np.sqrt((d.reshape(len(a),-1)**2).sum(1))/np.sqrt((a.reshape(len(a),-1)**2).sum(1))).max()
        out+= [f'alpha fro={fro:.2e} maxrel={np.abs(d/a).max():.2e}']
    out+=[f"u maxabs={np.abs(res.u.cpu().numpy()-g['u']).max():.2e} v maxabs={np.abs(res.v.cpu().numpy()-g['v']).max():.2e}"]
    print(*out, flush=True)
    if not np.array_equal(mm,g['mm_iters']): print('  mm', mm.tolist(), g['mm_iters'].tolist())
    print('  crit', res.criterions.cpu().numpy()[0][:4], g['criterions'][:4])
