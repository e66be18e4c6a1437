"""A small mesh scene built through the headless front-end: a room of quads, a textured table, a pyramid of triangles and a
sphere with smooth per-corner normals; one distant and one spot light.  Pure data (numpy); used by the command-line harness
(``python -m opencl_render_amd``) and by the front-end tests."""
import numpy as np

from . import frontend as F


def quad_box(lo, hi, material, inward=False, open_front=False):
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    pts = np.array([[x, y, z] for z in (lo[2], hi[2]) for y in (lo[1], hi[1]) for x in (lo[0], hi[0])], np.float32)
    quads = np.array([[0, 1, 3, 2], [4, 6, 7, 5], [0, 4, 5, 1], [2, 3, 7, 6], [0, 2, 6, 4], [1, 5, 7, 3]], np.int32)
    if open_front:
        quads = quads[1:]  # no face at z = lo: the camera looks in from outside
    if inward:
        quads = quads[:, ::-1].copy()
    uv = np.tile(np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32), (len(quads), 1, 1))
    return F.Mesh(points=pts, polygons=quads, corner_uv=uv, polygon_material=np.full(len(quads), material, np.int32))


def pyramid(base, apex, material):
    b = np.asarray(base, np.float32)
    pts = np.vstack([b, np.asarray(apex, np.float32)[None]])
    tris = np.array([[0, 1, 4, 4], [1, 2, 4, 4], [2, 3, 4, 4], [3, 0, 4, 4]], np.int32)  # c == d: triangles
    return F.Mesh(points=pts, polygons=tris, polygon_material=np.array([material, material, -1, material], np.int32))


def uv_sphere(centre, radius, material, rings=6, segs=10):
    pts, pol, nrm = [], [], []
    for i in range(rings + 1):
        th = np.pi * i / rings
        for j in range(segs):
            ph = 2 * np.pi * j / segs
            pts.append([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])
    pts = np.array(pts, np.float32)
    for i in range(rings):
        for j in range(segs):
            a, b = i * segs + j, i * segs + (j + 1) % segs
            c, d = (i + 1) * segs + (j + 1) % segs, (i + 1) * segs + j
            pol.append([a, b, c, d])
            nrm.append([pts[a] * 3.0, pts[b] * 0.5, pts[c] * 2.0, pts[d]])  # un-normalised on purpose: the front-end normalises
    world = (pts * np.float32(radius) + np.asarray(centre, np.float32)).astype(np.float32)
    return F.Mesh(points=world, polygons=np.array(pol, np.int32), corner_normals=np.array(nrm, np.float32),
                  polygon_material=np.full(len(pol), material, np.int32))


def materials():
    rng = np.random.Generator(np.random.PCG64(3))
    checker = np.zeros((8, 8, 3), np.uint8)
    checker[::2, ::2] = checker[1::2, 1::2] = (230, 220, 200)
    checker[::2, 1::2] = checker[1::2, ::2] = (60, 70, 90)
    return [
        dict(rgb=(0.8, 0.8, 0.75), brightness=1.0),                               # walls: colour from the material colour
        dict(color=checker, reflection=True),                                     # table: bitmap colour, default reflectance 0.2
        dict(rgb=(0.9, 0.3, 0.2), brightness=0.9, luminance=rng.integers(0, 40, (4, 4, 3)).astype(np.uint8)),
        dict(rgb=(0.7, 0.8, 1.0), transparency=np.full((1, 1, 3), 150, np.uint8), bump=rng.integers(0, 256, (6, 6, 3)).astype(np.uint8)),
    ]


def lights():
    return [dict(type=3, dir=(0.3, -0.8, 0.5), col=(1.0, 0.95, 0.9), brightness=0.9),
            dict(type=1, pos=(0.5, 1.6, 0.8), dir=(0, -1, 0), col=(0.6, 0.7, 1.0), brightness=0.7)]


def room_scene(width=160, height=120, samples=2):
    # The whole scene lies in front of the eye: the reference's camera lists project every vertex and have no clipping for
    # triangles that reach behind the camera (trianglelist.cpp:547 "TODO: Error will happen when a triangle is partially behind
    # the camera"), so the room is open towards the viewer.
    meshes = [quad_box((-2, 0, 0.2), (2, 2.5, 5), 0, inward=True, open_front=True), quad_box((-0.8, 0.6, 1.8), (0.8, 0.7, 3.0), 1),
              pyramid([(-1.6, 0, 2.2), (-1.0, 0, 2.2), (-1.0, 0, 2.8), (-1.6, 0, 2.8)], (-1.3, 0.9, 2.5), 2),
              uv_sphere((0.2, 1.05, 2.4), 0.35, 3)]
    return F.scene_from_meshes(meshes, materials(), lights(), position=(0.1, 1.3, -2.2), look_at=(0.0, 0.9, 2.5), up=(0, 1, 0),
                               fov=np.radians(60.0), width=width, height=height, samples=samples, name="room")
