import os, sys, copy
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from opencl_render_amd import raytrace as R, scene as S
import oracle_lib as O
mats = [dict(color=np.random.default_rng(4).integers(0, 256, (16, 16, 3)), reflection=(0, 0, 0), transparency=(0, 0, 0), bump=(0, 0, 0), luminance=(0, 0, 0))]
base = S.make_soup(256, 192, 9000, 0.05, seed=71, samples=2, materials=mats, random_uv=True)
R.build_lists(base)
texel = copy.copy(base); texel.textures = base.textures.copy(); texel.textures[:256, :3] = 255 - texel.textures[:256, :3]
want = O.oracle_render(texel, threads=os.cpu_count())
def bad(p): return [int((np.asarray(a).reshape(192,256) != w).sum()) for a, w in zip(p, want)]
R.lib().rtHipCacheClear()
ok, r, g, b = R.raytrace_all(1, texel); print("fresh build of the edited scene:", bad((r, g, b)))
ok, r, g, b = R.raytrace_all(1, texel); print("again (planned):", bad((r, g, b)))
R.lib().rtHipCacheClear()
ok, r0, g0, b0 = R.raytrace_all(1, base)
ok, r, g, b = R.raytrace_all(1, texel); print("after base, materials rebuilt:", bad((r, g, b)))
ok, r, g, b = R.raytrace_all(1, texel); print("again:", bad((r, g, b)))
got = R.render_resident(texel, 0); print("resident layer:", bad(got))
os.environ["RT_WF_ORDERED_FIRST"] = "0"
got = R.render_resident(texel, 0); print("resident layer, trace-planned:", bad(got))
