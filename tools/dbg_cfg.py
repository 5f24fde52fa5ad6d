import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from jpeg_amd import binding
from jpeg_amd.synth import CONFIGS, synth_image
name=sys.argv[1]
X,Y,nc,P,H,V,q,opt,safe,seed=CONFIGS[name]
img=synth_image(X,Y,nc,P,seed,safe)
jpgs,st=binding.encode([img],P=P,H=H,V=V,q=q,optimize=opt)
print(st,len(jpgs[0]))
d=binding.Decoder(); d.prepare(jpgs); out=d.run(); print(d.sync(), d.stats(0))
