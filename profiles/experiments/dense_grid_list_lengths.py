"""Development helper: how long the camera-ray lists of tests/test_gpu_parity.py::test_camera_ray_lists_of_many_nodes_and_several_rounds are (run on the GPU box against
the statistics build, which can dump the lists:  bash profiles/experiments/build.sh stats -DRTOW_STATS ;
RTOW_LIB_PATH=raytracing-in-one-weekend_amd/csrc/build/librtow_hip_stats.so python profiles/experiments/dense_grid_list_lengths.py).
Round 4, final kernel: 12 x 8 pixels, pinhole: 16 / 7 / 14 / 11 / 17 / 14 / 9 / 3 lists of 1 .. 8 nodes, 5 pixels without; 48 x 32 with the lens: 417 lists of 3 .. 8 nodes, 1 119 pixels
without - the test covers the second half of the lists, their continuation rounds and the walk of the camera rays that have none."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
rt = importlib.import_module("raytracing-in-one-weekend_amd")
S = rt.scenes
def scene(aperture):
    s = S.Scene("dense grid under wide pixels")
    rng = np.random.default_rng(3)
    for ix in range(7):
        for iy in range(5):
            for iz in range(2):
                s.add_sphere((-0.9 + 0.3 * ix + 0.02 * rng.random(), -0.6 + 0.3 * iy + 0.02 * rng.random(), -0.4 * iz), 0.11 + 0.03 * rng.random(),
                             S.lambertian((0.2 + 0.1 * ix, 0.3 + 0.1 * iy, 0.5)) if (ix + iy + iz) % 3 else S.metal((0.8, 0.8, 0.6), 0.1))
    s.add_sphere((0.0, -100.8, 0.0), 100.0, S.lambertian((0.5, 0.5, 0.5)))
    s.camera = {"position": [0.1, 0.05, 3.0], "target": [0.0, 0.0, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 40.0, "aperture": aperture}
    return s
def main():
    for ap in (0.0, 0.2):
        for w, h in ((6, 6), (12, 8), (48, 32)):
            os.environ["RTOW_DUMP_PRIMARY_LISTS"] = "/tmp/l.bin"
            s = scene(ap)
            with rt.Context(0) as ctx:
                ctx.upload_scene(s.desc())
                p = S.make_params(s, w, h, spp=2, trace_depth=4, focus=3.0)
                rt.sample_batch_host(ctx, p)
            a = np.fromfile("/tmp/l.bin", dtype=np.uint16).reshape(-1, 8)[: w * h]
            nolist = (a[:, 0] == 0xffff) & (a[:, 1] != 0xffff)
            cnt = (a != 0xffff).sum(axis=1)
            cnt[nolist] = 9
            print("aperture", ap, w, h, "nodes per list (9 = no list):", np.bincount(cnt, minlength=10).tolist())


if __name__ == "__main__":
    main()
