#!/usr/bin/env python3
'''GPU-box stress (not a test): 40 seeded random scenes (1..900 triangles beside the walls, random opaque materials, lights,
film sizes, spp and batch sizes) through the strict build and the four production kernels (LDS-resident over 4-wide and over binary nodes, binary gather, 4-wide gather); prints each kernel's distance
from the strict film and whether the LDS-resident and binary gather kernels agree bit for bit.  Round 3: all 40 agree bit for
bit between kernels, outliers <= 0.04 % (bound 0.5 %); the three rel-RMSE marks are single firefly pixels where the STRICT
build and the oracle differ by one libm-last-bit decision (seed 129 checked against the oracle).'''
import os, sys, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np
from ptina_amd import scenes, common
from ptina_amd.common import ctx, reset_all
from ptina_amd.things import FilmTable
from ptina_amd.tools.matrix import translate
from helpers import setup_engine, image_stats, FAST
bad=0
for seed in range(100, 140):
    rng=np.random.default_rng(seed)
    walls=scenes.cornell_walls()
    k=int(rng.integers(1, 900))
    c=rng.uniform([-1.6,0.3,-1.6],[1.6,3.4,1.2],(k,1,3))
    P=c+rng.normal(0,1,(k,3,3))*rng.uniform(0.03,0.7,(k,1,1))
    fn=np.cross(P[:,1]-P[:,0],P[:,2]-P[:,0]); fn/=np.linalg.norm(fn,axis=1,keepdims=True)+1e-30
    N=fn[:,None,:]+rng.normal(0,0.25,(k,3,3)); N/=np.linalg.norm(N,axis=2,keepdims=True)
    T=rng.uniform(0,1,(k,3,2)); nm=int(rng.integers(2,6)); M=rng.integers(3,3+nm,k).astype(np.int32)
    v,m=scenes._compose([walls,(P,N,T,M)])
    mats=list(scenes.WALL_MATERIALS)
    for _ in range(nm):
        mats.append(scenes.material(basecolor=tuple(rng.uniform(0,1,3)*(rng.random()>0.15)), metallic=float(rng.random()**2), roughness=float(rng.uniform(0.05,1)), specular=float(rng.random()), specularTint=float(rng.random()), subsurface=float(rng.random()*(rng.random()>0.5)), sheen=float(rng.random()*(rng.random()>0.5)), sheenTint=float(rng.random())))
    scene=(v,m,mats,[])
    rot=np.eye(4); rot[:3,:3]=[[1,0,0],[0,0,1],[0,-1,0]]
    lights=[]
    for _ in range(int(rng.integers(1,4))):
        pos=rng.uniform([-1.5,2.2,-1.5],[1.5,3.8,1.5])
        if rng.random()<0.5: lights.append((translate(list(pos))@rot, rng.uniform(4,20,3), float(rng.uniform(0.2,0.7)),'AREA'))
        else: lights.append((translate(list(pos)), rng.uniform(4,20,3), float(rng.uniform(0.05,0.4)),'POINT'))
    world=([float(x) for x in rng.uniform(0,0.4,3)]+[1.0],-1)
    nx,ny,spp=int(rng.integers(20,200)),int(rng.integers(20,160)),int(rng.integers(1,40))
    imgs={}
    kernels={}
    for name,mode,opts in (('strict','strict',{}),('lds4','fast',{}),('lds','fast',{'lds_wide':0}),('bin','fast',{'lds':0,'wide':0}),('wide','fast',{'lds':0})):
        reset_all()
        eng=setup_engine(scene,nx,ny,mode=mode,lights=lights,world=world)
        for a,b in opts.items(): ctx().set_option(a,b)
        ctx().set_option('batch', int(rng.integers(1,33)))
        eng.render(spp)
        raw=FilmTable().get_raw().reshape(nx,ny,4)
        assert np.all(raw[...,3]==spp), (seed,name)
        imgs[name]=FilmTable().get_image()
        kernels[name]=ctx().get_option('last_kernel')
    msg=[]
    for name in ('lds4','lds','bin','wide'):
        d,refn,rel=image_stats(imgs[name],imgs['strict'])
        out=float((d>FAST[0]*(1+refn)).mean())
        ok = out<=FAST[1]*2 and rel<=FAST[2]*2
        if not ok: bad+=1
        msg.append(f'{name} rel {rel:.1e} out {out*100:.2f}%'+('' if ok else ' <<<<'))
    same=np.array_equal(imgs['lds'].view(np.uint32),imgs['bin'].view(np.uint32))
    same4=float((imgs['lds4'].view(np.uint32)==imgs['lds'].view(np.uint32)).all(axis=-1).mean())
    print(seed,k+10,'tris',nx,ny,spp,'|',' | '.join(msg),'| lds==bin',same,'| lds4==lds on %.3f %% of the pixels'%(100*same4),'| kernels',kernels,flush=True)
print('bad',bad)
reset_all()
