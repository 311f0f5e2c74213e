import sys, os
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),'tests'), os.path.join(os.getcwd(),'tools')]
import numpy as np
import wgpu_3dgs_core_amd as gs
from oracle import binding as ob
import helpers, synth
dev=gs.Device(0); st=gs.Stream(dev)
for n,scale,op in ((120000,2.0,250),(120000,3.0,250),(120000,4.0,250),(60000,3.0,255)):
    g=synth.scene(n, first=4242); g["color"][:,3]=op; g["scale"]*=np.float32(scale)
    pod=gs.GaussianPod(gs.SH_NONE, gs.COV3D_ROT_SCALE); pods=pod.from_gaussian(g)
    ocam=helpers.default_camera(ob,640,360); cam=helpers.copy_camera(ocam, gs.Camera)
    gt=gs.gaussian_transform_pod(1.0,0,0,False,3.0); mt=gs.model_transform_pod((0,0,0),(0,0,0,1),(1,1,1))
    buf=gs.GaussiansBuffer.new_with_pods(dev,pod,pods)
    img=gs.Buffer(dev,size=640*360*16)
    r=gs.Renderer(dev); r.set_rounds(0); r.render(st,buf,gt,mt,cam,img.device_ptr()); fr=r.wait_frame()
    print(n,scale,op,"V",fr.visible,"D",fr.pairs)
    for k in (fr.visible//8, fr.visible//4, fr.visible//2):
        r2=gs.Renderer(dev); r2.set_rounds(1,k); r2.render(st,buf,gt,mt,cam,img.device_ptr()); f2=r2.wait_frame(); si=r2.sort_info()
        print("   K",si.round1,"rounds",si.rounds,"tiles_done",si.tiles_done,"of",40*23,"pairs",f2.pairs)
