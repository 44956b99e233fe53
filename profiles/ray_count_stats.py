import sys, importlib, numpy as np
sys.path.insert(0,'/root/repo')
rt = importlib.import_module("raytracing-in-one-weekend_amd")
scene = rt.scenes.cover_scene()
w,h=1920,1080; n=w*h
with rt.Context(0) as ctx:
    ctx.upload_scene(scene.desc())
    bufs=[rt.DeviceBuffer(ctx,n*k*4).zero() for k in (4,3,3,1)]
    diag=rt.DeviceBuffer(ctx,n*4).zero()
    for depth in (8,16,32,64):
        p=rt.scenes.make_params(scene,w,h,spp=256,trace_depth=depth)
        for rep in range(2):
            for b in bufs: b.zero()
            job=rt.SampleBatchJob(ctx,p)
            job.InputColor,job.InputNormal,job.InputAlbedo,job.InputSampleCountWeight=bufs
            job.OutputColor,job.OutputNormal,job.OutputAlbedo,job.OutputSampleCountWeight=bufs
            job.OutputDiagnostics=diag
            rt.lib.check(job.Schedule().Complete(),"x"); ctx.synchronize()
        ms=ctx.last_sample_kernel_ms()
        d=diag.download(np.float32,(n,))
        i=int(d.argmax())
        # per 8x8 tile (chunk) sums
        t=d.reshape(h//8,8,w//8,8).sum(axis=(1,3)) if h%8==0 else None
        print("depth",depth,"ms",round(ms,2),"rays total",d.sum(),"max/pixel",d.max(),"at",(i%w,i//w),"p99.9",np.percentile(d,99.9),"mean",d.mean(), "us per ray of max pixel if critical", round(ms*1e3/d.max(),2), "throughput-bound ms", round(d.sum()/23.9e6,2))
        top=np.sort(d)[-5:]; print("  top5", top, " pixels > half max:", int((d>d.max()/2).sum()))
