"""Measure the 16-bit paths against the reference fixtures (tolerance discovery)."""
import importlib, sys, os, numpy as np, torch
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests"); sys.path.insert(0, ROOT+"/oracle")
eng=importlib.import_module("gan-discovery-pso_amd.engine"); synth=importlib.import_module("gan-discovery-pso_amd.synth")
import pso_oracle as oracle
dev=torch.device("cuda:0")
g=np.load(ROOT+"/tests/golden/layers.npz")
for dt in ("f32","bf16","f16"):
  for tag,zd in (("z10",10),("z100",100)):
    gen=eng.PackedGenerator(synth.generator_state(zd,1,64,seed=11),dev,dtype=dt)
    ass=eng.PackedAssessor(synth.resnet_state("ResNet50",1,8,seed=12),dev,img_side=28,dtype=dt)
    ev=eng.Evaluator(gen,ass)
    z=torch.from_numpy(g[tag+"_z"]).cuda()
    img01,img,_=ev.images(z)
    ci=sorted(g[tag+"_iid"].tolist()).index(int(g[tag+"_label"]))
    fit,logits=ev.assess(img01,eng.FitnessSpec(class_index=ci),want_logits=True)
    print(dt,tag,"img",np.abs(img.cpu().numpy()-g[tag+"_img"][:,0]).max(),"img01",np.abs(img01.cpu().numpy()-g[tag+"_img01"][:,0]).max(),
          "logits",np.abs(logits.cpu().numpy()-g[tag+"_logits"]).max(), "of", np.abs(g[tag+"_logits"]).max(), "fit",np.abs(fit.cpu().numpy()-g[tag+"_fitness"]).max())
# larger sample vs oracle
for zd in (10,100):
    gsd=synth.to_torch(synth.generator_state(zd,1,64,seed=11)); csd=synth.to_torch(synth.resnet_state("ResNet50",1,8,seed=12))
    z=np.random.RandomState(3).standard_normal((256,zd))
    for ci in (1,6):
      want=oracle.Fitness(gsd,csd,ci)(z)
      for dt in ("f32","bf16","f16"):
        ev=eng.Evaluator(eng.PackedGenerator(synth.generator_state(zd,1,64,seed=11),dev,dtype=dt),eng.PackedAssessor(synth.resnet_state("ResNet50",1,8,seed=12),dev,img_side=28,dtype=dt))
        got,_,_=ev.fitness(torch.from_numpy(z).cuda(),eng.FitnessSpec(class_index=ci))
        d=np.abs(got.cpu().numpy()-want)
        print("zd",zd,"class",ci,dt,"max|dfit|",d.max(),"mean",d.mean(), "fit range", want.min(), want.max())
