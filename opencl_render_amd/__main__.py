"""Headless harness: ``python -m opencl_render_amd --scene room --width 640 --height 480 --samples 16 --out room.bmp``.

The counterpart of the plugin's dialog fields (image size, samples per pixel, device: reference ``source/render.cpp:174-186``)
and of the tail of ``parseAndRender`` (``render.cpp:1311-1397``): build a scene through the front-end, build both lists on the
GPU, render through the drop-in ``RaytraceAll`` and write the image the way the reference does (BMP in ``writebmp3s``'s layout,
or PPM).  Needs an MI355X: there is no CPU fallback.
"""
import argparse
import sys
import time

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m opencl_render_amd", description=__doc__.splitlines()[0])
    ap.add_argument("--scene", choices=["room", "soup"], default="room", help="demo room (meshes through the front-end) or a seeded triangle soup")
    ap.add_argument("--obj", help="render this Wavefront OBJ (its MTL libraries and PPM / BMP textures are read too) instead of a demo scene")
    ap.add_argument("--eye", type=float, nargs=3, default=None, help="--obj: camera position (default: in front of the model's bounding box)")
    ap.add_argument("--look-at", type=float, nargs=3, default=None, help="--obj: point the camera looks at (default: the bounding box's centre)")
    ap.add_argument("--fov", type=float, default=50.0, help="--obj: horizontal field of view in degrees")
    ap.add_argument("--light-dir", type=float, nargs=3, default=(0.3, -0.8, 0.5), help="--obj: direction of the one distant light")
    ap.add_argument("--width", type=int, default=1024)   # the dialog's defaults (render.cpp:176-182)
    ap.add_argument("--height", type=int, default=768)
    ap.add_argument("--samples", type=int, default=100)
    ap.add_argument("--triangles", type=int, default=100_000, help="soup only")
    ap.add_argument("--device", type=int, default=1, help="computationType: 1..N = HIP device, N+1 = all GPUs (tiled)")
    ap.add_argument("--out", default="img.bmp", help=".bmp or .ppm")
    ap.add_argument("--low-byte-compat", action="store_true", help="BMP only: keep the low byte of every u16 like the reference's writebmp3s")
    args = ap.parse_args(argv)

    from . import demo, frontend, raytrace, scene
    if raytrace.lib().rtHipDeviceCount() < 1:
        sys.exit("no HIP device visible: this library has no CPU fallback")
    names = raytrace.computation_type_names()
    if not (1 <= args.device < len(names)):
        sys.exit(f"--device {args.device}: choose 1..{len(names) - 1} ({names[1:]})")
    t0 = time.perf_counter()
    if args.obj:
        mesh, materials = frontend.read_obj(args.obj)
        lo, hi = mesh.points.min(axis=0), mesh.points.max(axis=0)
        centre, size = (lo + hi) / 2, float(np.linalg.norm(hi - lo)) or 1.0
        look_at = np.asarray(args.look_at, np.float32) if args.look_at else centre
        eye = np.asarray(args.eye, np.float32) if args.eye else centre + np.float32([0.35, 0.25, -1.0]) * size
        sc = frontend.scene_from_meshes([mesh], materials, [dict(type=scene.LIGHT_DISTANT, dir=tuple(args.light_dir))], eye, look_at, (0, 1, 0),
                                        np.radians(args.fov), args.width, args.height, samples=args.samples, name=args.obj)
    elif args.scene == "room":
        sc = demo.room_scene(args.width, args.height, samples=args.samples)
    else:
        sc = scene.make_soup(args.width, args.height, args.triangles, 0.02, samples=args.samples)
    t1 = time.perf_counter()
    cam_ms = raytrace.build_camera_list_device(sc, 0)
    grid_ms = raytrace.build_scene_grid_device(sc, 0)
    t2 = time.perf_counter()
    ok, r, g, b = raytrace.raytrace_all(args.device, sc)
    t3 = time.perf_counter()
    if not ok:
        sys.exit("RaytraceAll failed: " + raytrace.last_error())
    if args.out.lower().endswith(".ppm"):
        frontend.write_ppm(args.out, r, g, b)
    else:
        frontend.write_bmp(args.out, r, g, b, low_byte_compat=args.low_byte_compat)
    rays = args.width * args.height * args.samples
    print(f"{names[args.device]}: {sc.name}, {sc.triangle_count} triangles, {args.width}x{args.height}, {args.samples} samples/pixel -> {args.out}\n"
          f"  scene {1e3 * (t1 - t0):.0f} ms, lists on the device {1e3 * (t2 - t1):.0f} ms (kernels {cam_ms:.1f} + {grid_ms:.1f} ms), "
          f"RaytraceAll {1e3 * (t3 - t2):.0f} ms = {rays / (t3 - t2) / 1e6:.0f} M primary rays/s, lit pixels {float((np.asarray(r) > 0).mean()):.2f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
