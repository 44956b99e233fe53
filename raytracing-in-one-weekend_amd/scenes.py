"""Synthetic benchmark scenes for the sample-batch path (host side, float32 numpy).

The book-cover scene exists in the reference only as data
(`Assets/Scenes/Legacy/Final Scene (Book 1).asset`) plus a commented-out generator
(`Assets/Scripts/Unity/Raytracer.cs:1355-1506`, `CollectActiveEntities`).  This module restates that
generator - dart throwing driven by Unity.Mathematics.Random (xorshift32) - and emits the flat
`RtowSceneDesc` arrays of include/rtow.h.  It is input preparation for BOTH the product and the
oracle; it contains no part of the sample path itself.

Frozen build decisions:
 * fixed-sphere materials dangle in the asset (unresolved GUIDs, asset lines 33,48,63,78) -> the book's:
   ground lambert 0.5 grey, centre glass ior 1.5, left lambert (0.4,0.2,0.1), right metal (0.7,0.6,0.5) fuzz 0;
 * legacy material kinds map onto the reference's current model (RT/Material.cs:9-14):
     lambert      -> Standard{metallic 0, glossiness 0}
     metal(fuzz)  -> Standard{metallic 1, glossiness 1 - fuzz}
     glass(ior)   -> Dielectric{ior, glossiness 1, albedo 1}
 * the asset's 1000 tentatives give 4 + N random spheres with N != 482; BASELINE.json labels the scene
   "486-sphere", so `cover_scene()` keeps throwing darts from the same stream until exactly 486 entities
   exist and records how many tentatives that took (`tentatives_used`).
"""
import math

import ctypes as C

import numpy as np

from . import abi

f32 = np.float32
_M32 = 0xFFFFFFFF
UM_PI = f32(3.14159265)


class UnityRandom:
    """Unity.Mathematics.Random 1.2.5 (assumed published semantics; package not vendored in the reference).

    Random(seed): state = seed; NextState().   NextState(): returns the PRE-update state.
    NextFloat(): asfloat(0x3f800000 | (NextState() >> 9)) - 1.
    """

    def __init__(self, seed):
        self.state = seed & _M32
        self.next_state()

    def next_state(self):
        t = self.state
        s = t
        s ^= (s << 13) & _M32
        s ^= s >> 17
        s ^= (s << 5) & _M32
        self.state = s
        return t

    def next_float(self):
        bits = np.array([0x3F800000 | (self.next_state() >> 9)], dtype=np.uint32)
        return f32(bits.view(np.float32)[0] - f32(1.0))

    def next_float_range(self, lo, hi):
        lo, hi = f32(lo), f32(hi)
        return f32(f32(self.next_float() * f32(hi - lo)) + lo)

    def next_float3(self):
        x = self.next_float()
        y = self.next_float()
        z = self.next_float()
        return np.array([x, y, z], dtype=np.float32)

    def next_float3_range(self, lo, hi):
        lo = np.asarray(lo, dtype=np.float32)
        hi = np.asarray(hi, dtype=np.float32)
        return (self.next_float3() * (hi - lo) + lo).astype(np.float32)


# ---------------------------------------------------------------------------------------------------
# material helpers (legacy MaterialData.Lambertian / Metal / Dielectric -> RtowMaterial)
# ---------------------------------------------------------------------------------------------------
def _const_tex(v):
    if np.isscalar(v):
        v = (v, v, v)
    return abi.Texture(abi.TEXTURE_CONSTANT, abi.Float3(*[float(f32(c)) for c in v]), 0.0, 0, -1)


def _none_tex():
    return abi.Texture(abi.TEXTURE_NONE, abi.Float3(0, 0, 0), 0.0, 0, -1)


def image_tex(image_index, main_color=(1.0, 1.0, 1.0), channel=0):
    """TextureType.Image (RT/Texture.cs:80-89,126-135): texel / 255 * MainColor; `image_index` into Scene.images (-1: null pointer)."""
    return abi.Texture(abi.TEXTURE_IMAGE, abi.Float3(*[float(f32(c)) for c in main_color]), 0.0, channel, image_index)


def scalar_tex(value):
    """TextureType.ConstantScalar (RT/Texture.cs:58-59,103-104)."""
    return abi.Texture(abi.TEXTURE_CONSTANT_SCALAR, abi.Float3(0, 0, 0), float(f32(value)), 0, -1)


def lambertian(color):
    return abi.Material(abi.MATERIAL_STANDARD, _const_tex(color), _const_tex(0.0), _none_tex(), _const_tex(0.0), 0.0)


def metal(color, fuzz):
    gloss = f32(f32(1.0) - f32(fuzz))
    return abi.Material(abi.MATERIAL_STANDARD, _const_tex(color), _const_tex(gloss), _none_tex(), _const_tex(1.0), 0.0)


def dielectric(ior, gloss=1.0, albedo=1.0):
    return abi.Material(abi.MATERIAL_DIELECTRIC, _const_tex(albedo), _const_tex(gloss), _none_tex(), _none_tex(), float(f32(ior)))


def volume(color, density):
    """ProbabilisticVolume (RT/Material.cs:14,40-42,49-65,163-168): isotropic scattering medium of constant density."""
    return abi.Material(abi.MATERIAL_PROBABILISTIC_VOLUME, _const_tex(color), _none_tex(), _none_tex(), _none_tex(), float(f32(density)))


def standard(color, metallic, gloss, emission=None):
    em = _none_tex() if emission is None else _const_tex(emission)
    return abi.Material(abi.MATERIAL_STANDARD, _const_tex(color), _const_tex(gloss), em, _const_tex(metallic), 0.0)


class Scene:
    """Flat scene: parallel Python lists that `desc()` packs into an RtowSceneDesc."""

    def __init__(self, name):
        self.name = name
        self.positions = []      # float32[3]
        self.radii = []          # float32 (signed)
        self.moving = []         # bool
        self.dest_offsets = []   # float32[3]
        self.time_ranges = []    # (t0, t1)
        self.material_index = []
        self.exclude_from_overlap = []
        self.types = []          # abi.ENTITY_*
        self.rotations = []      # quaternion (x, y, z, w)
        self.sizes = []          # float32[3]: sphere (r,-,-), rect (sx,sy,-), box (sx,sy,sz)
        self.tri_index = []      # index into self.triangles, -1 for non-triangles
        self.triangles = []      # abi.Triangle payloads
        self.materials = []      # abi.Material
        self.images = []         # uint8 arrays [H, W, C] referenced by Image textures (RtowTexture.imageIndex)
        self.camera = {}
        self.sky_bottom = (1.0, 1.0, 1.0)
        self.sky_top = (0.5, 0.7, 1.0)
        self.meta = {}
        self._keepalive = None

    @property
    def entity_count(self):
        return len(self.radii)

    def _add(self, etype, pos, size, material, rotation, moving, dest_offset, time_range, exclude, tri=-1):
        if isinstance(material, int):
            mi = material
        else:
            self.materials.append(material)
            mi = len(self.materials) - 1
        self.types.append(etype)
        self.positions.append(np.asarray(pos, dtype=np.float32))
        self.sizes.append(np.asarray(size, dtype=np.float32))
        self.radii.append(f32(size[0]))
        self.rotations.append(tuple(float(f32(c)) for c in rotation))
        self.moving.append(bool(moving))
        self.dest_offsets.append(np.asarray(dest_offset, dtype=np.float32))
        self.time_ranges.append((f32(time_range[0]), f32(time_range[1])))
        self.material_index.append(mi)
        self.exclude_from_overlap.append(bool(exclude))
        self.tri_index.append(tri)

    def add_sphere(self, pos, radius, material, moving=False, dest_offset=(0, 0, 0), time_range=(0, 0), exclude=False, rotation=(0, 0, 0, 1)):
        self._add(abi.ENTITY_SPHERE, pos, (radius, 0, 0), material, rotation, moving, dest_offset, time_range, exclude)

    def add_rect(self, pos, size_xy, material, rotation=(0, 0, 0, 1), moving=False, dest_offset=(0, 0, 0), time_range=(0, 0)):
        """Rect in the entity's XY plane, facing +Z (RT/EntityTypes/Rect.cs); only hit by rays with local direction.z < 0."""
        self._add(abi.ENTITY_RECT, pos, (size_xy[0], size_xy[1], 0), material, rotation, moving, dest_offset, time_range, True)

    def add_box(self, pos, size, material, rotation=(0, 0, 0, 1), moving=False, dest_offset=(0, 0, 0), time_range=(0, 0)):
        self._add(abi.ENTITY_BOX, pos, size, material, rotation, moving, dest_offset, time_range, True)

    def add_triangle(self, v1, v2, v3, material, normals=None, uvs=((0, 0), (1, 0), (0, 1)), moving=False, dest_offset=(0, 0, 0), time_range=(0, 0)):
        """World-space triangle; Triangle ctor of RT/EntityTypes/Triangle.cs:15-30 (face normal when `normals` is None).
        The entity carries the `default` RigidTransform (zero quaternion), as AddMeshRuntimeEntitiesJob creates it (:83-85)."""
        v1, v2, v3 = (np.asarray(v, dtype=np.float32) for v in (v1, v2, v3))
        d0, d1 = (v3 - v1).astype(np.float32), (v2 - v1).astype(np.float32)
        if normals is None:
            fn = _normalize(_cross(d1, d0))
            ns = (fn, fn, fn)
        else:
            ns = tuple(_normalize(np.asarray(n, dtype=np.float32)) for n in normals)
        t = abi.Triangle()
        for k, v in enumerate((d0, d1, v1)):
            t.data[k] = abi.Float3(*[float(c) for c in v])
        for k, v in enumerate(ns):
            t.normals[k] = abi.Float3(*[float(c) for c in v])
        for k, v in enumerate(uvs):
            t.textureCoordinates[k] = abi.Float2(float(v[0]), float(v[1]))
        self.triangles.append(t)
        self._add(abi.ENTITY_TRIANGLE, (0, 0, 0), (0, 0, 0), material, (0, 0, 0, 0), moving, dest_offset, time_range, True, tri=len(self.triangles) - 1)

    def desc(self, max_bvh_depth=32):
        n = self.entity_count
        ents = (abi.Entity * n)()
        for i in range(n):
            e = ents[i]
            e.type = self.types[i]
            e.moving = 1 if self.moving[i] else 0
            e.rotation = abi.Float4(*self.rotations[i])  # quaternion.Euler(0,0,0) = (0,0,0,1) for the book scenes
            e.position = abi.Float3(*[float(c) for c in self.positions[i]])
            e.destinationOffset = abi.Float3(*[float(c) for c in self.dest_offsets[i]])
            e.timeRange = abi.Float2(float(self.time_ranges[i][0]), float(self.time_ranges[i][1]))
            e.materialIndex = self.material_index[i]
            e.size = abi.Float3(*[float(c) for c in self.sizes[i]])
            e.contentIndex = max(self.tri_index[i], 0)
        mats = (abi.Material * len(self.materials))(*self.materials)
        tris = (abi.Triangle * max(len(self.triangles), 1))(*self.triangles)
        imgs = (abi.Image * max(len(self.images), 1))()
        pix = [np.ascontiguousarray(im, dtype=np.uint8) for im in self.images]
        for k, im in enumerate(pix):
            imgs[k] = abi.Image(im.shape[1], im.shape[0], im.shape[2], im.ctypes.data)
        d = abi.SceneDesc(ents, n, mats, len(self.materials), max_bvh_depth, tris if self.triangles else None, len(self.triangles),
                          imgs if self.images else None, len(self.images))
        self._keepalive = (ents, mats, tris, imgs, pix)
        return d

    # -- reproducible serialisation for tests/golden ------------------------------------------------
    def to_dict(self):
        def tex(t):
            return [int(t.type), float(t.mainColor.x), float(t.mainColor.y), float(t.mainColor.z), float(t.parameter), int(t.scalarValueChannel)]

        return {
            "name": self.name,
            "positions": [[float(c) for c in p] for p in self.positions],
            "radii": [float(r) for r in self.radii],
            "moving": [int(m) for m in self.moving],
            "dest_offsets": [[float(c) for c in p] for p in self.dest_offsets],
            "time_ranges": [[float(a), float(b)] for a, b in self.time_ranges],
            "material_index": list(self.material_index),
            "materials": [[int(m.type), tex(m.albedo), tex(m.glossiness), tex(m.emission), tex(m.metallic), float(m.parameter)] for m in self.materials],
            "camera": self.camera,
            "sky_bottom": list(self.sky_bottom),
            "sky_top": list(self.sky_top),
            "meta": self.meta,
        }

    @staticmethod
    def from_dict(d):
        s = Scene(d["name"])

        def tex(t):
            return abi.Texture(t[0], abi.Float3(t[1], t[2], t[3]), t[4], t[5], -1)

        s.materials = [abi.Material(m[0], tex(m[1]), tex(m[2]), tex(m[3]), tex(m[4]), m[5]) for m in d["materials"]]
        s.positions = [np.asarray(p, dtype=np.float32) for p in d["positions"]]
        s.radii = [f32(r) for r in d["radii"]]
        s.moving = [bool(m) for m in d["moving"]]
        s.dest_offsets = [np.asarray(p, dtype=np.float32) for p in d["dest_offsets"]]
        s.time_ranges = [(f32(a), f32(b)) for a, b in d["time_ranges"]]
        s.material_index = list(d["material_index"])
        s.exclude_from_overlap = [False] * len(s.radii)
        s.types = [abi.ENTITY_SPHERE] * len(s.radii)
        s.rotations = [(0.0, 0.0, 0.0, 1.0)] * len(s.radii)
        s.sizes = [np.array([r, 0, 0], dtype=np.float32) for r in s.radii]
        s.tri_index = [-1] * len(s.radii)
        s.camera = d["camera"]
        s.sky_bottom = tuple(d["sky_bottom"])
        s.sky_top = tuple(d["sky_top"])
        s.meta = d.get("meta", {})
        return s


# ---------------------------------------------------------------------------------------------------
# RandomEntityGroup dart throwing (UNITY/Raytracer.cs:1357-1505, commented-out generator)
# ---------------------------------------------------------------------------------------------------
def _throw_darts(scene, rng, *, tentative_count, spread, offset, radius_range, min_distance,
                 chances, diffuse_range, double_sample, metal_range, fuzz_range, ior_range,
                 movement_chance, movement_lo, movement_hi, stop_at_entity_count=None):
    lambert_c, metal_c, diel_c, light_c = [f32(c) for c in chances]
    total = f32(f32(f32(lambert_c + metal_c) + diel_c) + light_c)          # :1371
    metal_c = f32(metal_c + lambert_c)                                      # :1372-1374
    diel_c = f32(diel_c + metal_c)
    light_c = f32(light_c + diel_c)
    p_lambert, p_metal, p_diel = f32(lambert_c / total), f32(metal_c / total), f32(diel_c / total)  # :1375-1378

    offset = np.asarray(offset, dtype=np.float32)
    half = np.asarray(spread, dtype=np.float32) / f32(2)
    used = 0

    def get_material():                                                    # :1363-1413
        v = rng.next_float()
        if v < p_lambert:
            color = rng.next_float3_range(diffuse_range[0], diffuse_range[1])
            if double_sample:
                color = (color * rng.next_float3_range(diffuse_range[0], diffuse_range[1])).astype(np.float32)
            return lambertian(color)
        if v < p_metal:
            color = rng.next_float3_range(metal_range[0], metal_range[1])
            fuzz = rng.next_float_range(fuzz_range[0], fuzz_range[1])
            return metal(color, fuzz)
        if v < p_diel:
            return dielectric(rng.next_float_range(ior_range[0], ior_range[1]))
        return None

    for _ in range(tentative_count):
        if stop_at_entity_count is not None and scene.entity_count >= stop_at_entity_count:
            break
        used += 1
        center = rng.next_float3_range(-half, half)                        # :1456-1458
        center = (center + offset).astype(np.float32)                      # :1460
        radius = rng.next_float_range(radius_range[0], radius_range[1])    # :1462

        # AnyOverlap (:1416-1419): distance(x.Position, center) < x.Radius + radius + MinDistance
        if scene.entity_count:
            pos = np.stack(scene.positions).astype(np.float32)
            rad = np.asarray(scene.radii, dtype=np.float32)
            excl = np.asarray(scene.exclude_from_overlap, dtype=bool)
            d = (center[None, :] - pos).astype(np.float32)
            dist = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(np.float32)).astype(np.float32)
            limit = ((rad + radius).astype(np.float32) + f32(min_distance)).astype(np.float32)
            if np.any(~excl & (dist < limit)):
                continue

        # GetEntity (:1421-1451)
        moving = bool(rng.next_float() < f32(movement_chance))
        position = ((center - offset).astype(np.float32) + offset).astype(np.float32)  # rotate(identity, c - o) + o
        material = get_material()
        dest = np.zeros(3, dtype=np.float32)
        time_range = (0.0, 0.0)
        if moving:
            dest = rng.next_float3_range(movement_lo, movement_hi)
            time_range = (0.0, 1.0)
        if material is None:
            continue
        scene.add_sphere(position, radius, material, moving=moving, dest_offset=dest, time_range=time_range)
    return used


_BOOK_CAMERA = {"position": [12.3, 1.98, -2.99], "target": [11.342338, 1.8494737, -2.7333953], "up": [0.0, 1.0, 0.0], "vfov": 20.0}


def _book_fixed_spheres(scene):
    # `Final Scene (Book 1).asset`:19-83
    scene.add_sphere((0, -1000, 0), 1000, lambertian((0.5, 0.5, 0.5)), exclude=True)
    scene.add_sphere((0, 1, 0), 1, dielectric(1.5))
    scene.add_sphere((-4, 1, 0), 1, lambertian((0.4, 0.2, 0.1)))
    scene.add_sphere((4, 1, 0), 1, metal((0.7, 0.6, 0.5), 0.0))


def cover_scene(target_entity_count=486, max_tentatives=4000):
    """Book-1 cover scene (`Final Scene (Book 1).asset`): 4 fixed spheres + dart-thrown r=0.2 spheres, seed 700."""
    s = Scene("cover")
    _book_fixed_spheres(s)
    rng = UnityRandom(700)                                                  # asset :84
    used = _throw_darts(
        s, rng, tentative_count=max_tentatives, spread=(22, 0, 22), offset=(0, 0.2, 0), radius_range=(0.2, 0.2),
        min_distance=0.15, chances=(0.8, 0.15, 0.05, 0.0), diffuse_range=((0, 0, 0), (1, 1, 1)), double_sample=True,
        metal_range=((0.5, 0.5, 0.5), (1, 1, 1)), fuzz_range=(0, 0.5), ior_range=(1.5, 1.5),
        movement_chance=0.0, movement_lo=(0, 0, 0), movement_hi=(0, 0, 0), stop_at_entity_count=target_entity_count)
    s.camera = dict(_BOOK_CAMERA, aperture=0.0)
    s.meta = {"seed": 700, "tentatives_used": used, "asset_tentative_count": 1000, "target_entity_count": target_entity_count}
    return s


def moving_scene(tentative_count=1000):
    """`Random With Movement (Book 2).asset`: cover layout, MovementChance 0.8, Y offset in [0, 0.5], aperture 0.05."""
    s = Scene("moving")
    _book_fixed_spheres(s)
    rng = UnityRandom(700)
    used = _throw_darts(
        s, rng, tentative_count=tentative_count, spread=(22, 0, 22), offset=(0, 0.2, 0), radius_range=(0.2, 0.2),
        min_distance=0.15, chances=(0.8, 0.15, 0.05, 0.0), diffuse_range=((0, 0, 0), (1, 1, 1)), double_sample=True,
        metal_range=((0.5, 0.5, 0.5), (1, 1, 1)), fuzz_range=(0, 0.5), ior_range=(1.5, 1.5),
        movement_chance=0.8, movement_lo=(0, 0, 0), movement_hi=(0, 0.5, 0))
    s.camera = dict(_BOOK_CAMERA, aperture=0.05)
    s.meta = {"seed": 700, "tentatives_used": used}
    return s


def stress_scene(count=10000, seed=10000, spread=100.0, max_tentatives=60000):
    """Synthetic deep-BVH stress scene (BASELINE.json config 4): `count` r in [0.05, 0.2] spheres dart-thrown on a
    spread x spread area with the cover scene's material mix.  Nothing like it exists in the reference."""
    s = Scene("stress%d" % count)
    _book_fixed_spheres(s)
    rng = UnityRandom(seed)
    used = _throw_darts(
        s, rng, tentative_count=max_tentatives, spread=(spread, 0, spread), offset=(0, 0.2, 0), radius_range=(0.05, 0.2),
        min_distance=0.05, chances=(0.8, 0.15, 0.05, 0.0), diffuse_range=((0, 0, 0), (1, 1, 1)), double_sample=True,
        metal_range=((0.5, 0.5, 0.5), (1, 1, 1)), fuzz_range=(0, 0.5), ior_range=(1.5, 1.5),
        movement_chance=0.0, movement_lo=(0, 0, 0), movement_hi=(0, 0, 0), stop_at_entity_count=count)
    s.camera = dict(_BOOK_CAMERA, aperture=0.0)
    s.meta = {"seed": seed, "tentatives_used": used}
    return s


def quat_axis_angle(axis, degrees):
    """quaternion.AxisAngle(axis, radians) = (sin(a/2) * axis, cos(a/2)) in float32."""
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    h = math.radians(degrees) / 2
    q = np.array([a[0] * math.sin(h), a[1] * math.sin(h), a[2] * math.sin(h), math.cos(h)], dtype=np.float32)
    return tuple(float(c) for c in q)


def mixed_scene():
    """Cornell-like room exercising every primitive kind and the general entity transform: rects (walls, light), rotated
    and moving boxes, a rotated sphere, a small triangle mesh (pyramid, face normals + one smooth-shaded face)."""
    s = Scene("mixed")
    white, red, green = lambertian((0.73, 0.73, 0.73)), lambertian((0.65, 0.05, 0.05)), lambertian((0.12, 0.45, 0.15))
    light = standard((0.0, 0.0, 0.0), 0.0, 0.0, emission=(7.0, 7.0, 7.0))
    up90, dn90 = quat_axis_angle((1, 0, 0), -90), quat_axis_angle((1, 0, 0), 90)
    s.add_rect((0, 0, -2), (4, 4), white)                                          # back wall, faces +Z
    s.add_rect((0, -2, 0), (4, 4), white, rotation=up90)                           # floor, faces +Y
    s.add_rect((0, 2, 0), (4, 4), white, rotation=dn90)                            # ceiling, faces -Y
    s.add_rect((-2, 0, 0), (4, 4), red, rotation=quat_axis_angle((0, 1, 0), 90))   # left wall, faces +X
    s.add_rect((2, 0, 0), (4, 4), green, rotation=quat_axis_angle((0, 1, 0), -90)) # right wall, faces -X
    s.add_rect((0, 1.99, 0), (1.2, 1.2), light, rotation=dn90)                     # area light
    s.add_box((-0.7, -1.2, -0.6), (1.1, 1.6, 1.1), white, rotation=quat_axis_angle((0, 1, 0), 20))
    s.add_box((0.8, -1.5, 0.3), (1.0, 1.0, 1.0), metal((0.8, 0.85, 0.9), 0.1), rotation=quat_axis_angle((0, 1, 0), -17),
              moving=True, dest_offset=(0.0, 0.3, 0.0), time_range=(0.0, 1.0))
    s.add_sphere((0.7, -0.6, 0.3), 0.4, dielectric(1.5), rotation=quat_axis_angle((1, 1, 0), 33))
    s.add_sphere((-0.7, 0.0, -0.6), 0.35, metal((0.9, 0.6, 0.2), 0.0), rotation=quat_axis_angle((0, 0, 1), 75),
                 moving=True, dest_offset=(0.2, 0.0, 0.0), time_range=(0.2, 0.9))
    apex = (0.0, -0.9, 1.0)
    base = [(-0.4, -2.0, 0.6), (0.4, -2.0, 0.6), (0.4, -2.0, 1.4), (-0.4, -2.0, 1.4)]
    blue = lambertian((0.2, 0.3, 0.8))
    for k in range(4):
        a, b = base[k], base[(k + 1) % 4]
        if k == 0:
            s.add_triangle(a, apex, b, blue, normals=((-0.5, 0.3, -1), (0, 1, 0), (0.5, 0.3, -1)))   # smooth-shaded face
        else:
            s.add_triangle(a, apex, b, blue)
    s.camera = {"position": [0.0, 0.0, 6.5], "target": [0.0, 0.0, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 38.0, "aperture": 0.0}
    s.sky_bottom, s.sky_top = (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)
    return s


def volume_scene():
    """`Cornell With Volumes (Book 2)`-like set-up: smoke and fog boxes, a fog sphere with a solid sphere inside it, a glass
    sphere partly inside the fog, and a camera ray path that starts INSIDE a big thin haze sphere (containment probe)."""
    s = Scene("volumes")
    white, red, green = lambertian((0.73, 0.73, 0.73)), lambertian((0.65, 0.05, 0.05)), lambertian((0.12, 0.45, 0.15))
    light = standard((0.0, 0.0, 0.0), 0.0, 0.0, emission=(7.0, 7.0, 7.0))
    up90, dn90 = quat_axis_angle((1, 0, 0), -90), quat_axis_angle((1, 0, 0), 90)
    s.add_rect((0, 0, -2), (4, 4), white)
    s.add_rect((0, -2, 0), (4, 4), white, rotation=up90)
    s.add_rect((0, 2, 0), (4, 4), white, rotation=dn90)
    s.add_rect((-2, 0, 0), (4, 4), red, rotation=quat_axis_angle((0, 1, 0), 90))
    s.add_rect((2, 0, 0), (4, 4), green, rotation=quat_axis_angle((0, 1, 0), -90))
    s.add_rect((0, 1.99, 0), (1.6, 1.6), light, rotation=dn90)
    s.add_box((-0.8, -1.2, -0.5), (1.1, 1.6, 1.1), volume((0.05, 0.05, 0.05), 1.8), rotation=quat_axis_angle((0, 1, 0), 20))     # smoke
    s.add_box((0.8, -1.5, 0.2), (1.0, 1.0, 1.0), volume((0.95, 0.95, 0.95), 2.5), rotation=quat_axis_angle((0, 1, 0), -17))      # fog
    s.add_sphere((0.0, 0.6, -0.2), 0.6, volume((0.3, 0.5, 0.9), 3.0))                                                          # fog ball ...
    s.add_sphere((0.0, 0.6, -0.2), 0.25, metal((0.9, 0.8, 0.3), 0.05))                                                         # ... with a solid core
    s.add_sphere((0.9, -0.6, 0.9), 0.35, dielectric(1.5))
    s.add_sphere((0.0, 0.0, 4.0), 3.5, volume((1.0, 1.0, 1.0), 0.04))                                                          # haze around the camera
    s.camera = {"position": [0.0, 0.0, 6.5], "target": [0.0, 0.0, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 38.0, "aperture": 0.0}
    s.sky_bottom, s.sky_top = (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)
    return s


def volume_tie_scene():
    """Axis-aligned fog boxes standing ON the floor and touching each other and a glass slab: many rays meet two or three
    surfaces at bit-identical distances, so the result depends on the reference's (unstable) hit-sort tie order."""
    s = Scene("volume_ties")
    white = lambertian((0.8, 0.8, 0.8))
    light = standard((0.0, 0.0, 0.0), 0.0, 0.0, emission=(4.0, 4.0, 4.0))
    up90, dn90 = quat_axis_angle((1, 0, 0), -90), quat_axis_angle((1, 0, 0), 90)
    s.add_rect((0, 0, 0), (8, 8), white, rotation=up90)                                  # floor y = 0
    s.add_rect((0, 0, -2), (8, 4), white)                                                # back wall z = -2 (its lower half is below the floor)
    s.add_rect((0, 3.0, 0), (3, 3), light, rotation=dn90)
    fog_a, fog_b = volume((0.9, 0.6, 0.3), 1.5), volume((0.3, 0.6, 0.9), 2.5)
    s.add_box((-0.5, 0.5, -1.5), (1, 1, 1), fog_a)                                       # bottom on the floor, back on the wall
    s.add_box((0.5, 0.5, -1.5), (1, 1, 1), fog_b)                                        # shares the x = 0 face with fog_a
    s.add_box((0.0, 0.25, -0.5), (2, 0.5, 1), fog_a)                                     # same medium, touching both from the front
    s.add_box((1.5, 0.5, -1.5), (1, 1, 1), dielectric(1.5))                              # glass block sharing the x = 1 face with fog_b
    s.add_rect((0.0, 1.0, -1.5), (2, 1), standard((0.9, 0.9, 0.9), 1.0, 0.9), rotation=up90)   # a mirror lid exactly on top of both boxes
    s.camera = {"position": [0.3, 2.2, 4.0], "target": [0.2, 0.4, -1.0], "up": [0.0, 1.0, 0.0], "vfov": 40.0, "aperture": 0.0}
    s.sky_bottom, s.sky_top = (0.3, 0.3, 0.3), (0.2, 0.3, 0.5)
    return s


def volume_stack_scene(slabs=10, thickness=0.5):
    """Ten thin fog slabs stacked face to face along the view axis, in front of a wall lying in the last slab's back face: a camera ray
    collects 21 hits (more than the 16 up to which the reference's sort is an insertion sort) of which ten pairs are at bit-identical
    distances (camera and faces on dyadic coordinates), so the result depends on the partition steps of the unstable introsort.
    More and thinner slabs (48 x 0.125) make hit lists of ~100 entries: the reference's list grows on the heap, the library's spills."""
    s = Scene("volume_stack")
    white = lambertian((0.8, 0.8, 0.8))
    up90 = quat_axis_angle((1, 0, 0), -90)
    s.add_rect((0, -1.5, 0), (12, 12), white, rotation=up90)
    s.add_rect((0, 0, -2.5), (3, 3), standard((0.0, 0.0, 0.0), 0.0, 0.0, emission=(3.0, 2.5, 2.0)))     # wall in the z = -2.5 face of the last slab
    fogs = [volume((0.9, 0.5, 0.3), 0.12), volume((0.3, 0.6, 0.9), 0.2), volume((0.5, 0.9, 0.4), 0.08)]
    order = [k for k in (3, 7, 0, 9, 12, 4, 1, 10, 8, 5, 11, 2, 6) if k < slabs]          # not in depth order
    if slabs > 13:
        order += [int(k) for k in np.random.default_rng(11).permutation(np.arange(13, slabs))]
    for k in order:
        s.add_box((0.0, 0.0, -2.5 + thickness * (k + 0.5)), (3, 3, thickness), fogs[k % 3])
    s.add_sphere((0.5, -0.25, 0.25), 0.5, dielectric(1.5))                                 # something solid inside the stack
    s.camera = {"position": [0.25, 0.5, 6.0], "target": [0.0, 0.0, -2.0], "up": [0.0, 1.0, 0.0], "vfov": 35.0, "aperture": 0.0}
    s.sky_bottom, s.sky_top = (0.4, 0.4, 0.4), (0.3, 0.4, 0.7)
    return s


def coplanar_scene():
    """No volumes: decals lying exactly IN the plane of a wall / the floor and boxes sharing faces - the nearest hit is a tie
    between two entities for many pixels, decided by the reference's hit-list order (leaf order of its tree)."""
    s = Scene("coplanar")
    up90 = quat_axis_angle((1, 0, 0), -90)
    s.add_rect((0, 0, 0), (10, 10), lambertian((0.7, 0.7, 0.7)), rotation=up90)          # floor
    s.add_rect((0, 2, -3), (10, 4), lambertian((0.6, 0.2, 0.2)))                          # wall
    s.add_rect((-1.5, 2, -3), (2, 2), metal((0.9, 0.9, 0.9), 0.0))                        # mirror decal in the wall's plane
    s.add_rect((1.5, 2, -3), (2, 2), standard((0.0, 0.0, 0.0), 0.0, 0.0, emission=(3.0, 2.5, 2.0)))   # light panel in the wall's plane
    s.add_rect((1.5, 2, -3), (1, 1), lambertian((0.1, 0.6, 0.1)))                         # ... and a decal on the decal
    s.add_rect((0, 0, 1), (3, 3), lambertian((0.2, 0.3, 0.8)), rotation=up90)             # rug in the floor's plane
    s.add_rect((0.5, 0, 1.5), (1, 1), metal((0.8, 0.7, 0.3), 0.2), rotation=up90)         # tile on the rug
    s.add_box((-2.5, 0.5, 0.0), (1, 1, 1), dielectric(1.5))                               # two glass blocks sharing a face
    s.add_box((-1.5, 0.5, 0.0), (1, 1, 1), dielectric(1.3))
    s.add_sphere((2.5, 0.5, 0.5), 0.5, lambertian((0.8, 0.5, 0.2)))
    s.add_sphere((2.5, 0.5, 0.5), 0.5, metal((0.9, 0.9, 0.9), 0.1))                       # the same sphere twice
    s.camera = {"position": [0.5, 2.5, 6.0], "target": [0.0, 1.0, -1.0], "up": [0.0, 1.0, 0.0], "vfov": 45.0, "aperture": 0.0}
    return s


def twin_spheres_scene(moving=False):
    """Sphere-only scene (the SPHERES / SPHERES_MOTION kernels) where spheres coincide exactly: twins and triplets with different
    materials, added in different orders, so the nearest hit is a tie decided by the reference tree's leaf order."""
    s = Scene("twin_spheres_moving" if moving else "twin_spheres")
    s.add_sphere((0, -100.5, 0), 100, lambertian((0.6, 0.6, 0.6)))
    mats = [lambertian((0.8, 0.2, 0.2)), metal((0.9, 0.9, 0.9), 0.0), dielectric(1.5), standard((0.1, 0.1, 0.1), 0.0, 0.0, emission=(2.0, 1.5, 1.0)),
            lambertian((0.1, 0.7, 0.2)), metal((0.8, 0.6, 0.2), 0.4)]
    rng = np.random.default_rng(5)
    k = 0
    for gx in range(-3, 4):
        for gz in range(-2, 3):
            pos = (gx * 1.1 + float(rng.uniform(-0.2, 0.2)), 0.0 + float(rng.uniform(0.0, 0.3)), gz * 1.1 + float(rng.uniform(-0.2, 0.2)))
            r = float(rng.uniform(0.3, 0.5))
            copies = 1 + (k % 3)                                   # single, twin, triplet
            mv = dict(moving=True, dest_offset=(0.0, 0.4, 0.1), time_range=(0.0, 1.0)) if (moving and k % 2 == 0) else {}
            for c in range(copies):
                s.add_sphere(pos, r, mats[(k * 5 + c * (1 + k % 4)) % len(mats)], **mv)
            k += 1
    s.camera = {"position": [0.5, 3.0, 7.0], "target": [0.0, 0.2, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 40.0, "aperture": 0.05 if moving else 0.0}
    return s


def decal_stack_scene(layers=20):
    """No volumes, no duplicate primitives: a wall with decals IN its plane (ties between different surfaces) in front of `layers` more panes
    along the view axis, so that rays through the decals have more than 16 hits - the case in which the order the reference's unstable sort
    leaves the tied nearest hits in is no longer the leaf order and only the whole procedure reproduces it."""
    s = Scene("decal_stack")
    mats = [lambertian((0.8, 0.2, 0.2)), lambertian((0.2, 0.8, 0.2)), metal((0.9, 0.9, 0.9), 0.1), standard((0.1, 0.1, 0.1), 0.0, 0.0, emission=(2.0, 1.5, 1.0)),
            lambertian((0.2, 0.3, 0.9)), dielectric(1.5)]
    order = [int(k) for k in np.random.default_rng(31).permutation(layers)]
    for k in order[: layers // 2]:
        s.add_rect((0.0, 0.0, -0.25 * (k + 1)), (6.0 - 0.125 * k, 6.0 - 0.125 * k), mats[k % len(mats)])
    s.add_rect((0.0, 0.0, 0.0), (4.0, 4.0), mats[0])                                   # the wall ...
    s.add_rect((-1.0, 0.5, 0.0), (1.5, 1.5), mats[2])                                  # ... a mirror decal in its plane
    s.add_rect((1.0, -0.5, 0.0), (1.5, 1.0), mats[3])                                  # ... a light panel in its plane
    s.add_rect((1.0, -0.5, 0.0), (0.5, 0.5), mats[1])                                  # ... and a decal on the panel
    s.add_box((-1.0, -1.25, 0.25), (1.0, 1.0, 0.5), mats[4])                           # a box standing on the wall: its back face lies in the plane too
    for k in order[layers // 2:]:
        s.add_rect((0.0, 0.0, -0.25 * (k + 1)), (6.0 - 0.125 * k, 6.0 - 0.125 * k), mats[k % len(mats)])
    s.camera = {"position": [0.25, 0.5, 7.0], "target": [0.0, 0.0, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 40.0, "aperture": 0.0}
    return s


def twin_row_scene(count=30, moving=False):
    """A row of `count` coinciding sphere pairs along the view axis: rays near the axis pass through every one of them (up to 2 x count
    hits) and the nearest hit is always a tie, so the exact-tie procedure sorts hit lists far longer than a lane's own 24 entries."""
    s = Scene("twin_row_moving" if moving else "twin_row")
    s.add_sphere((0, -100.5, 0), 100, lambertian((0.6, 0.6, 0.6)))
    mats = [lambertian((0.8, 0.2, 0.2)), metal((0.9, 0.9, 0.9), 0.0), dielectric(1.5), standard((0.1, 0.1, 0.1), 0.0, 0.0, emission=(2.0, 1.5, 1.0)),
            lambertian((0.1, 0.7, 0.2)), metal((0.8, 0.6, 0.2), 0.4)]
    order = [int(k) for k in np.random.default_rng(23).permutation(count)]                 # not in depth order
    for k in order:
        mv = dict(moving=True, dest_offset=(0.0, 0.25, 0.0), time_range=(0.0, 1.0)) if (moving and k % 3 == 0) else {}
        for c in range(2):
            s.add_sphere((0.0, 0.25, 4.0 - 0.75 * k), 0.375, mats[(k * 5 + c * (1 + k % 4)) % len(mats)], **mv)
    s.camera = {"position": [0.0, 0.25, 8.0], "target": [0.0, 0.25, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 12.0, "aperture": 0.0}
    return s


def _quad(s, p00, p10, p11, p01, material, uv0=(0.0, 0.0), uv1=(1.0, 1.0)):
    """Two triangles p00-p10-p11 / p00-p11-p01 with the texture coordinates of a [uv0, uv1] rectangle."""
    (u0, v0), (u1, v1) = uv0, uv1
    s.add_triangle(p00, p10, p11, material, uvs=((u0, v0), (u1, v0), (u1, v1)))
    s.add_triangle(p00, p11, p01, material, uvs=((u0, v0), (u1, v1), (u0, v1)))


def triangle_layers_scene(layers=20, duplicate=False):
    """An all-triangle scene (what the reference's live host produces) built to tie: `layers` parallel walls of two triangles each across the whole view, so that
    a ray has more than 16 hits, and in front of them two DIFFERENT triangles that meet every ray through the smaller one at bit-identical distance: the second is the
    first scaled by two about their common first vertex (Data = {2 e0, 2 e1, v0}: the determinant and the numerator of HitTests.Hit(Triangle) both pick up an exact factor
    of four, RT/HitTests.cs:116-139), in another material - a tie that only the reference's whole hit-list procedure settles.
    `duplicate`: the same front triangle twice on top (the scene then keeps the exact-tie kernels for every pixel: SceneLayout.tieWatchOk = 0)."""
    s = Scene("triangle_layers_dup" if duplicate else "triangle_layers")
    mats = [lambertian((0.8, 0.25, 0.2)), metal((0.9, 0.9, 0.9), 0.0), dielectric(1.5), standard((0.05, 0.05, 0.05), 0.0, 0.0, emission=(2.0, 1.6, 1.1)),
            lambertian((0.15, 0.65, 0.25)), metal((0.8, 0.6, 0.2), 0.3), dielectric(1.33)]
    order = [int(k) for k in np.random.default_rng(29).permutation(layers)]               # not in depth order
    for k in order:
        z = -0.25 * (k + 1)
        m = mats[(2 + k * 3) % len(mats)]
        if k % 2: _quad(s, (-6, -4, z), (6, -4, z), (6, 4, z), (-6, 4, z), m)
        else:
            s.add_triangle((-6, -4, z), (6, -4, z), (-6, 4, z), m)
            s.add_triangle((6, -4, z), (6, 4, z), (-6, 4, z), m)
    # the tied pair: v1 = (-4, -3, 0), small = v1 + {(4, 0, 0), (0, 3.5, 0)}, large = v1 + {(8, 0, 0), (0, 7, 0)}; and a second pair mirrored about the view axis
    s.add_triangle((-4.0, -3.0, 0.0), (0.0, -3.0, 0.0), (-4.0, 0.5, 0.0), mats[0])
    s.add_triangle((-4.0, -3.0, 0.0), (4.0, -3.0, 0.0), (-4.0, 4.0, 0.0), mats[3])
    s.add_triangle((4.0, 3.0, 0.0), (0.0, 3.0, 0.0), (4.0, -0.5, 0.0), mats[1])
    s.add_triangle((4.0, 3.0, 0.0), (-4.0, 3.0, 0.0), (4.0, -4.0, 0.0), mats[4])
    if duplicate: s.add_triangle((-4.0, -3.0, 0.0), (0.0, -3.0, 0.0), (-4.0, 0.5, 0.0), mats[5])
    s.camera = {"position": [0.3, 0.2, 7.0], "target": [0.0, 0.0, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 40.0, "aperture": 0.0}
    return s


def textured_scene(triangles_only=False):
    """Triangle meshes with Image textures on every texture slot (albedo, emission, glossiness, metallic; Standard and Dielectric),
    a constant-scalar texture, a null image pointer, and image-textured spheres / rects (whose texture coordinates are always 0).
    The reference's texture assets are not in the mount: the images are procedural."""
    s = Scene("textured")
    rng = np.random.default_rng(21)
    yy, xx = np.mgrid[0:32, 0:48]
    checker = np.where(((xx // 6) + (yy // 4)) % 2 == 0, 230, 40).astype(np.uint8)
    albedo = np.stack([checker, (xx * 5) % 256, (yy * 7) % 256], axis=-1).astype(np.uint8)                 # 32 x 48, RGB24
    emissive = np.zeros((16, 16, 4), dtype=np.uint8)                                                         # RGBA32: a few hot texels
    emissive[4:7, 9:12, :3] = (255, 200, 120)
    emissive[12, 2, :3] = (90, 90, 255)
    params = rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)                                                 # gloss in .g, metal in .b
    s.images = [albedo, emissive, params]
    wall = abi.Material(abi.MATERIAL_STANDARD, image_tex(0), _const_tex(0.0), _none_tex(), _const_tex(0.0), 0.0)
    lamp = abi.Material(abi.MATERIAL_STANDARD, image_tex(0, (0.2, 0.2, 0.2)), _const_tex(0.0), image_tex(1, (6.0, 6.0, 6.0)), _const_tex(0.0), 0.0)
    shiny = abi.Material(abi.MATERIAL_STANDARD, image_tex(0, (0.9, 0.8, 0.7)), image_tex(2, (1.0, 1.0, 1.0), channel=1), _none_tex(), image_tex(2, (1.0, 1.0, 0.9), channel=2), 0.0)
    frosted = abi.Material(abi.MATERIAL_DIELECTRIC, _const_tex((1.0, 1.0, 1.0)), image_tex(2, (1.0, 1.0, 1.0), channel=0), _none_tex(), _none_tex(), 1.5)
    missing = abi.Material(abi.MATERIAL_STANDARD, image_tex(-1), scalar_tex(0.3), _none_tex(), scalar_tex(0.5), 0.0)     # null ImagePointer -> albedo 0
    ball = abi.Material(abi.MATERIAL_STANDARD, image_tex(0, (1.0, 0.5, 0.5)), _const_tex(0.2), _none_tex(), _const_tex(0.1), 0.0)
    plain = lambertian((0.6, 0.6, 0.6))
    _quad(s, (-3, 0, -3), (3, 0, -3), (3, 0, 3), (-3, 0, 3), wall, uv1=(3.0 / 3.0, 1.0))                      # floor
    _quad(s, (-3, 0, -3), (-3, 4, -3), (3, 4, -3), (3, 0, -3), shiny, uv0=(0.0, 0.0), uv1=(0.999, 0.999))    # back wall
    _quad(s, (-3, 0, 3), (-3, 4, 3), (-3, 4, -3), (-3, 0, -3), lamp)                                          # left wall with hot texels
    _quad(s, (0.2, 0.3, 0.5), (1.8, 0.3, 0.5), (1.8, 1.9, 0.2), (0.2, 1.9, 0.2), frosted, uv0=(0.1, 0.2), uv1=(0.9, 0.8))   # glass pane
    _quad(s, (-2.2, 0.01, 0.5), (-1.0, 0.01, 0.5), (-1.0, 0.01, 1.7), (-2.2, 0.01, 1.7), missing)
    _quad(s, (-1.5, 3.99, -1.5), (1.5, 3.99, -1.5), (1.5, 3.99, 1.5), (-1.5, 3.99, 1.5), standard((0, 0, 0), 0.0, 0.0, emission=(5.0, 5.0, 5.0)))
    if triangles_only:                                  # 14 triangles and nothing else: the all-triangle textured kind, without the exact-tie resolver
        _quad(s, (2.99, 0.5, -1.0), (2.99, 0.5, 1.0), (2.99, 2.5, 1.0), (2.99, 2.5, -1.0), ball, uv0=(0.2, 0.1), uv1=(0.7, 0.9))
    else:
        s.add_sphere((-1.2, 0.6, -0.8), 0.6, ball)                                                            # texture coordinates (0, 0): one texel
        s.add_rect((2.99, 1.5, 0.0), (2.0, 2.0), wall, rotation=quat_axis_angle((0, 1, 0), -90))
        s.add_sphere((0.9, 0.45, 1.6), 0.45, plain)
    s.camera = {"position": [0.3, 1.8, 6.5], "target": [0.0, 1.3, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 48.0, "aperture": 0.0}
    s.sky_bottom, s.sky_top = (0.05, 0.05, 0.08), (0.1, 0.15, 0.3)
    return s


def textured_mesh_scene(subdivisions=1):
    """mesh_scene with image textures on its materials (albedo map on the first icosphere and the floor, a gloss / metal map on the third): an
    all-triangle textured scene of more than 16 entities (the exact-tie kernels of the all-triangle textured kind).  Icosphere vertices carry
    spherical texture coordinates."""
    s = mesh_scene(subdivisions)
    s.name = "textured mesh"
    yy, xx = np.mgrid[0:16, 0:24]
    albedo = np.stack([np.where(((xx // 3) + (yy // 2)) % 2 == 0, 220, 60), (xx * 9) % 256, (yy * 13) % 256], axis=-1).astype(np.uint8)
    params = np.random.default_rng(5).integers(0, 256, (8, 8, 3), dtype=np.uint8)
    s.images = [albedo, params]
    s.materials[0] = abi.Material(abi.MATERIAL_STANDARD, image_tex(0, (0.9, 0.9, 0.9)), _const_tex(0.0), _none_tex(), _const_tex(0.0), 0.0)
    s.materials[2] = abi.Material(abi.MATERIAL_STANDARD, _const_tex((0.8, 0.8, 0.9)), image_tex(1, (1.0, 1.0, 1.0), channel=1), _none_tex(), image_tex(1, (1.0, 1.0, 1.0), channel=2), 0.0)
    s.materials[3] = abi.Material(abi.MATERIAL_STANDARD, image_tex(0, (0.6, 0.6, 0.6)), _const_tex(0.0), _none_tex(), _const_tex(0.0), 0.0)
    for t in s.triangles:                                # texture coordinates from the vertex normals (spherical map); the floor keeps its quad coordinates
        if abs(t.normals[0].y) > 0.999 and abs(t.normals[1].y) > 0.999 and abs(t.normals[2].y) > 0.999:
            continue
        for k in range(3):
            nx, ny, nz = t.normals[k].x, t.normals[k].y, t.normals[k].z
            t.textureCoordinates[k] = abi.Float2(float(f32(0.5 + math.atan2(nz, nx) / (2 * math.pi))), float(f32(0.5 + 0.5 * ny)))
    return s


def textured_volume_scene():
    """`volume_scene` with Image textures: a textured floor mesh, an image-textured (uv = 0) wall, and fog whose albedo is an image."""
    s = volume_scene()
    s.name = "textured_volumes"
    rng = np.random.default_rng(5)
    s.images = [rng.integers(30, 256, (16, 16, 3), dtype=np.uint8), np.full((2, 2, 3), (200, 220, 255), dtype=np.uint8)]
    tiles = abi.Material(abi.MATERIAL_STANDARD, image_tex(0), _const_tex(0.3), _none_tex(), _const_tex(0.0), 0.0)
    _quad(s, (-2, -1.99, -2), (2, -1.99, -2), (2, -1.99, 2), (-2, -1.99, 2), tiles, uv1=(2.0, 2.0))      # a tiled mesh just above the floor rect
    s.materials[s.material_index[0]].albedo = image_tex(0, (0.73, 0.73, 0.73))                              # back wall rect: texel (0, 0) only
    for i, m in enumerate(s.materials):
        if m.type == abi.MATERIAL_PROBABILISTIC_VOLUME:
            m.albedo = image_tex(1, (m.albedo.mainColor.x, m.albedo.mainColor.y, m.albedo.mainColor.z))       # fog colour from an image
            break
    return s


def mesh_scene(subdivisions=3):
    """A triangle-mesh scene like the reference's live ones (UNITY/MeshData.cs feeds `Triangle` entities): three icospheres
    (20 * 4^subdivisions triangles each, smooth normals) of different materials on a two-triangle floor."""
    s = Scene("mesh")
    t = (1.0 + 5.0 ** 0.5) / 2.0
    verts = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [np.array(v, dtype=np.float64) / np.linalg.norm(v) for v in verts]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
             (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(subdivisions):
        cache, out = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            out += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = out
    for centre, radius, mat in (((-2.2, 1.0, 0.0), 1.0, lambertian((0.7, 0.3, 0.3))), ((0.0, 1.0, 0.0), 1.0, dielectric(1.5)), ((2.2, 1.0, 0.0), 1.0, metal((0.8, 0.8, 0.9), 0.05))):
        s.materials.append(mat)
        mi = len(s.materials) - 1
        c = np.array(centre)
        for a, b, cc in faces:
            s.add_triangle(c + radius * verts[a], c + radius * verts[b], c + radius * verts[cc], mi, normals=(verts[a], verts[b], verts[cc]))
    _quad(s, (-30, 0, -30), (30, 0, -30), (30, 0, 30), (-30, 0, 30), lambertian((0.5, 0.5, 0.5)))
    s.camera = {"position": [0.0, 2.2, 7.5], "target": [0.0, 0.9, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 35.0, "aperture": 0.0}
    return s


def _icosphere(subdivisions):
    """Unit icosphere: vertices [V, 3] float64 and faces [F, 3] int (F = 20 * 4^subdivisions)."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    verts = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [np.array(v, dtype=np.float64) / np.linalg.norm(v) for v in verts]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
             (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(subdivisions):
        cache, out = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            out += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = out
    return np.asarray(verts), np.asarray(faces, dtype=np.int64)


_ENTITY_DTYPE = np.dtype([("type", "<i4"), ("moving", "<i4"), ("rotation", "<f4", 4), ("position", "<f4", 3), ("destinationOffset", "<f4", 3), ("timeRange", "<f4", 2),
                          ("materialIndex", "<i4"), ("size", "<f4", 3), ("contentIndex", "<i4")])


class BulkScene(Scene):
    """A scene whose entities are kept as packed arrays (one RtowEntity / RtowTriangle record each) instead of per-entity Python lists: what a host
    that turns every mesh triangle into an entity hands over (UNITY/Raytracer.cs:1193-1198,1290-1300) - hundreds of thousands of records."""

    def __init__(self, name, entities, triangles, materials):
        super().__init__(name)
        assert entities.dtype == _ENTITY_DTYPE and _ENTITY_DTYPE.itemsize == C.sizeof(abi.Entity)
        assert triangles.dtype == np.float32 and triangles.ndim == 2 and triangles.shape[1] * 4 == C.sizeof(abi.Triangle)
        self.entities = np.ascontiguousarray(entities)
        self.triangle_records = np.ascontiguousarray(triangles)
        self.materials = list(materials)

    @property
    def entity_count(self):
        return len(self.entities)

    def desc(self, max_bvh_depth=32):
        n = len(self.entities)
        ents = (abi.Entity * n).from_buffer(self.entities)
        tris = (abi.Triangle * len(self.triangle_records)).from_buffer(self.triangle_records)
        mats = (abi.Material * len(self.materials))(*self.materials)
        d = abi.SceneDesc(ents, n, mats, len(self.materials), max_bvh_depth, tris, len(self.triangle_records), None, 0)
        self._keepalive = (ents, mats, tris)
        return d


def mesh_grid_scene(grid=(14, 14), subdivisions=3, spacing=2.4, radius=1.0):
    """The reference's own kind of test scene (UNITY/GridGenerator.cs:78-159): a grid of sphere MESHES whose material parameters blend across the
    grid - here icospheres of 20 * 4^subdivisions smooth-shaded triangles, metallic growing along one axis and glossiness along the other, on a
    two-triangle floor.  Every triangle is an entity (UNITY/Raytracer.cs:1193-1198): 14 x 14 x 1280 + 2 = 250 882 of them by default, far beyond
    the 65 535 that 16-bit candidate codes can name."""
    verts, faces = _icosphere(subdivisions)
    gx, gy = grid
    per = len(faces)
    total = gx * gy * per + 2
    tri = np.zeros((total, 24), np.float32)
    ent = np.zeros(total, _ENTITY_DTYPE)
    ent["type"] = abi.ENTITY_TRIANGLE
    ent["contentIndex"] = np.arange(total)
    materials = []
    v = verts.astype(np.float32)
    k = 0
    for j in range(gy):
        for i in range(gx):
            hs = 0.0 if gx == 1 else i / (gx - 1)
            vs = 0.0 if gy == 1 else j / (gy - 1)
            centre = np.array([(i - (gx - 1) / 2) * spacing, radius, -(j * spacing)], np.float32)
            materials.append(standard((0.9 - 0.5 * vs, 0.35 + 0.4 * hs, 0.3 + 0.5 * vs), float(f32(hs)), float(f32(0.15 + 0.85 * vs))))
            p = (centre[None, :] + f32(radius) * v).astype(np.float32)                    # world-space vertices of this sphere
            a, b, c = p[faces[:, 0]], p[faces[:, 1]], p[faces[:, 2]]
            block = tri[k:k + per]
            block[:, 0:3] = c - a                                                            # Data: v2 - v0, v1 - v0, v0 (RT/EntityTypes/Triangle.cs:15-30)
            block[:, 3:6] = b - a
            block[:, 6:9] = a
            for q, col in enumerate((0, 1, 2)):                                              # smooth normals = the unit sphere's directions, normalised like the ctor does
                nn = v[faces[:, col]]
                d = (nn[:, 0] * nn[:, 0] + nn[:, 1] * nn[:, 1]).astype(np.float32) + (nn[:, 2] * nn[:, 2]).astype(np.float32)
                block[:, 9 + 3 * q:12 + 3 * q] = ((np.float32(1.0) / np.sqrt(d, dtype=np.float32))[:, None] * nn).astype(np.float32)
            block[:, 18:24] = np.array([0, 0, 1, 0, 0, 1], np.float32)
            ent["materialIndex"][k:k + per] = len(materials) - 1
            k += per
    # floor: two triangles with face normals
    materials.append(lambertian((0.5, 0.5, 0.5)))
    ext = max(gx, gy) * spacing + 20.0
    quad = [np.array(q, np.float32) for q in ((-ext, 0, ext), (ext, 0, ext), (ext, 0, -ext - gy * spacing), (-ext, 0, -ext - gy * spacing))]
    for (a, b, c) in ((quad[0], quad[1], quad[2]), (quad[0], quad[2], quad[3])):
        d0, d1 = (c - a).astype(np.float32), (b - a).astype(np.float32)
        fn = _normalize(_cross(d1, d0))
        tri[k, 0:3], tri[k, 3:6], tri[k, 6:9] = d0, d1, a
        tri[k, 9:12] = tri[k, 12:15] = tri[k, 15:18] = fn
        tri[k, 18:24] = np.array([0, 0, 1, 0, 0, 1], np.float32)
        ent["materialIndex"][k] = len(materials) - 1
        k += 1
    assert k == total
    s = BulkScene("mesh_grid", ent, tri, materials)
    depth = gy * spacing
    s.camera = {"position": [0.0, 0.35 * depth + 3.0, 0.45 * depth + 6.0], "target": [0.0, 0.5, -0.45 * depth], "up": [0.0, 1.0, 0.0], "vfov": 42.0, "aperture": 0.0}
    s.meta = {"triangles": int(total), "grid": [gx, gy], "subdivisions": subdivisions, "focus": float(np.linalg.norm(np.array(s.camera["position"]) - np.array(s.camera["target"])))}
    return s


def mesh_grid_fog_scene(grid=(8, 8), subdivisions=3):
    """A grid of sphere meshes (mesh_grid_scene) with ProbabilisticVolume hulls standing among them: a fog ball around one mesh, a haze box across a
    row, a thin haze sphere around the camera (the containment probe runs for every camera ray).  One fog volume makes a triangle-mesh scene a
    VOLUME scene - every hit of a ray is kept and sorted - and 8 x 8 x 1280 + 2 + 3 = 81 925 entities put it beyond 16-bit candidate codes."""
    s = mesh_grid_scene(grid, subdivisions)
    gx, gy = grid
    spacing, radius = 2.4, 1.0
    extra = np.zeros(3, _ENTITY_DTYPE)
    extra["rotation"] = (0.0, 0.0, 0.0, 1.0)
    base = len(s.materials)
    s.materials += [volume((0.3, 0.5, 0.9), 1.2), volume((0.9, 0.9, 0.9), 0.25), volume((1.0, 1.0, 1.0), 0.02)]
    cx = ((gx // 2) - (gx - 1) / 2) * spacing
    extra[0]["type"], extra[0]["position"], extra[0]["size"], extra[0]["materialIndex"] = abi.ENTITY_SPHERE, (cx, radius, -(1 * spacing)), (1.6, 0, 0), base
    extra[1]["type"], extra[1]["position"], extra[1]["size"], extra[1]["materialIndex"] = abi.ENTITY_BOX, (0.0, 1.25, -(3 * spacing)), (gx * spacing, 2.5, 2.2), base + 1
    cam = np.array(s.camera["position"], np.float32)
    extra[2]["type"], extra[2]["position"], extra[2]["size"], extra[2]["materialIndex"] = abi.ENTITY_SPHERE, tuple(cam), (4.0, 0, 0), base + 2
    s.entities = np.ascontiguousarray(np.concatenate([s.entities, extra]))
    s.name = "mesh_grid_fog"
    s.meta = dict(s.meta, entities=int(len(s.entities)))
    return s


def tiny_scene():
    """Five spheres, one of each material branch + a negative-radius hollow glass shell; for fast unit tests."""
    s = Scene("tiny")
    s.add_sphere((0, -100.5, -1), 100, lambertian((0.8, 0.8, 0.0)))
    s.add_sphere((0, 0, -1), 0.5, lambertian((0.1, 0.2, 0.5)))
    s.add_sphere((-1, 0, -1), 0.5, dielectric(1.5))
    s.add_sphere((-1, 0, -1), -0.45, dielectric(1.5))
    s.add_sphere((1, 0, -1), 0.5, metal((0.8, 0.6, 0.2), 0.3))
    s.add_sphere((0.3, 0.9, -1.2), 0.3, standard((0.9, 0.3, 0.3), 0.4, 0.6, emission=(0.2, 0.1, 0.0)))
    s.add_sphere((-0.4, 0.6, -0.6), 0.15, metal((0.9, 0.9, 0.9), 0.0), moving=True, dest_offset=(0.0, 0.3, 0.1), time_range=(0.0, 1.0))
    s.camera = {"position": [-2.0, 2.0, 1.0], "target": [0.0, 0.0, -1.0], "up": [0.0, 1.0, 0.0], "vfov": 40.0, "aperture": 0.1}
    return s


# ---------------------------------------------------------------------------------------------------
# camera: RT/View.cs:16-36 + the auto-focus probe of UNITY/Raytracer.cs:608-609
# ---------------------------------------------------------------------------------------------------
def _normalize(v):
    v = v.astype(np.float32)
    d = f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]))
    return (f32(f32(1.0) / np.sqrt(d, dtype=np.float32)) * v).astype(np.float32)


def _cross(a, b):
    return np.array([f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(a[2] * b[0]) - f32(a[0] * b[2]), f32(a[0] * b[1]) - f32(a[1] * b[0])], dtype=np.float32)


def focus_distance(scene, origin, direction):
    """Nearest sphere hit along the view axis (HitWorld, UNITY/Raytracer.cs:608-609,1353) in float32 numpy."""
    o = np.asarray(origin, dtype=np.float32)
    d = np.asarray(direction, dtype=np.float32)
    sph = [i for i, t in enumerate(scene.types) if t == abi.ENTITY_SPHERE]
    if not sph:
        return 1.0
    pos = np.stack([scene.positions[i] for i in sph]).astype(np.float32)
    rad = np.asarray([scene.radii[i] for i in sph], dtype=np.float32)
    oc = (o[None, :] - pos).astype(np.float32)
    a = f32(d @ d)
    b = (oc @ d).astype(np.float32)
    c = ((oc * oc).sum(axis=1).astype(np.float32) - rad * rad).astype(np.float32)
    disc = (b * b - a * c).astype(np.float32)
    best = None
    for i in np.nonzero(disc > 0)[0]:
        sq = np.sqrt(disc[i], dtype=np.float32)
        for t in (f32((-b[i] - sq) / a), f32((-b[i] + sq) / a)):
            if t > 0:
                if best is None or t < best:
                    best = t
                break
    return float(best) if best is not None else 1.0


def make_view(scene, width, height, focus=None):
    """View ctor (RT/View.cs:16-36) -> abi.View.  aspect = width / height (TargetCamera.aspect)."""
    cam = scene.camera
    origin = np.asarray(cam["position"], dtype=np.float32)
    look_at = np.asarray(cam["target"], dtype=np.float32)
    up = np.asarray(cam["up"], dtype=np.float32)
    aperture = f32(cam.get("aperture", 0.0))
    if focus is None:
        fwd = _normalize(look_at - origin)
        focus = focus_distance(scene, origin, fwd)
    focus = f32(focus)
    aspect = f32(f32(width) / f32(height))

    lens_radius = f32(aperture / f32(2))
    theta = f32(f32(f32(cam["vfov"]) * UM_PI) / f32(180))
    half_height = f32(math.tan(float(f32(theta / f32(2)))))
    half_width = f32(aspect * half_height)

    forward = _normalize(origin - look_at)
    right = _normalize(_cross(forward, up))
    up_v = _cross(right, forward)

    llc = ((f32(half_width * focus) * -right).astype(np.float32) + (f32(half_height * focus) * -up_v).astype(np.float32)).astype(np.float32)
    llc = (llc + (focus * -forward).astype(np.float32)).astype(np.float32)
    horizontal = (f32(f32(f32(2) * half_width) * focus) * right).astype(np.float32)
    vertical = (f32(f32(f32(2) * half_height) * focus) * up_v).astype(np.float32)

    def v3(a):
        return abi.Float3(float(a[0]), float(a[1]), float(a[2]))

    return abi.View(v3(origin), v3(llc), v3(horizontal), v3(vertical), v3(forward), v3(up_v), v3(right), float(lens_radius))


class NoiseTextures:
    """Stand-ins for the host's noise textures (the reference's blue-noise assets are not in the mount; any texel values exercise the
    samplers): a set of `count` square half4 textures for BlueNoise, and the five byte-texture sets of SpatioTemporalBlueNoise."""

    def __init__(self, row_stride=16, count=2, seed=11):
        rng = np.random.default_rng(seed)
        n = count * row_stride * row_stride
        self.row_stride, self.count = row_stride, count
        self.blue = rng.random((n, 4), dtype=np.float32).astype(np.float16)                # half4, values in [0, 1)
        self.scalar = rng.integers(0, 256, n, dtype=np.uint8)
        self.vector2 = rng.integers(0, 256, (n, 3), dtype=np.uint8)
        self.unit_vector2 = rng.integers(0, 256, (n, 3), dtype=np.uint8)
        self.unit_vector3 = rng.integers(0, 256, (n, 3), dtype=np.uint8)
        # cosine-weighted unit vectors, stored the way the reference decodes them: (r, b, g) / 256 * 2 - 1 = (x, y, z), y up
        u, v = rng.random(n), rng.random(n)
        r, th = np.sqrt(u), 2 * np.pi * v
        x, y, z = r * np.cos(th), np.sqrt(1 - u), r * np.sin(th)
        enc = lambda a: np.clip(np.floor((a + 1) / 2 * 256), 0, 255).astype(np.uint8)
        self.cosine_unit_vector3 = np.stack([enc(x), enc(z), enc(y), np.full(n, 255, np.uint8)], axis=1)

    def blue_desc(self):
        return abi.BlueNoiseDesc(self.row_stride, self.count, self.blue.ctypes.data)

    def stb_desc(self):
        return abi.StbNoiseDesc(self.row_stride, self.count, self.scalar.ctypes.data, self.vector2.ctypes.data, self.cosine_unit_vector3.ctypes.data,
                                self.unit_vector2.ctypes.data, self.unit_vector3.ctypes.data)


class SkyCubemap:
    """Six faces (+X -X +Y -Y +Z -Z, contiguous) of a sky cube in the layout `Cubemap` expects (RT/Texture.cs:141-211)."""

    def __init__(self, faces, channel_type):
        self.faces = np.ascontiguousarray(faces)            # [6, H, W, C] float16 (C = 4: R16G16B16A16_SFloat) or uint8 (C = 3 / 4)
        assert self.faces.ndim == 4 and self.faces.shape[0] == 6
        self.channel_type = channel_type

    def desc(self):
        _, h, w, c = self.faces.shape
        return abi.CubemapDesc(w, h, self.channel_type, c * self.faces.itemsize, self.faces.ctypes.data)


def synthetic_sky(size=64, half=True, seed=3):
    """A procedural HDR sky (no HDRI asset travels with the reference): smooth gradient + a hot sun + per-texel noise so that
    neighbouring texels differ; half floats like the reference's only accepted format, or bytes for its UnsignedByte decode path."""
    rng = np.random.default_rng(seed)
    faces = np.zeros((6, size, size, 4), dtype=np.float32)
    axes = [(0, 1), (0, -1), (1, 1), (1, -1), (2, 1), (2, -1)]
    t = (np.arange(size, dtype=np.float32) + 0.5) / size * 2 - 1
    uu, vv = np.meshgrid(t, t)                              # uv of the texel centre: column = u, row = v
    sun = np.array([0.5, 0.7, -0.5], dtype=np.float32)
    sun /= np.linalg.norm(sun)
    for f, (axis, sign) in enumerate(axes):
        d = np.zeros((size, size, 3), dtype=np.float32)     # inverse of the uv mapping of RT/Texture.cs:184-190
        if axis == 0:
            d[..., 0], d[..., 2], d[..., 1] = sign, -sign * uu, -vv
        elif axis == 1:
            d[..., 1], d[..., 0], d[..., 2] = sign, uu, sign * vv
        else:
            d[..., 2], d[..., 0], d[..., 1] = sign, sign * uu, -vv
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        up = 0.5 * (d[..., 1] + 1)
        col = (1 - up)[..., None] * np.array([0.9, 0.8, 0.7], np.float32) + up[..., None] * np.array([0.3, 0.5, 1.0], np.float32)
        col += 40.0 * np.maximum(d @ sun, 0)[..., None] ** 200 * np.array([1.0, 0.9, 0.7], np.float32)
        col *= 0.9 + 0.2 * rng.random((size, size, 1), dtype=np.float32)
        faces[f, ..., :3] = col
        faces[f, ..., 3] = 1
    if half:
        return SkyCubemap(faces.astype(np.float16), abi.CUBEMAP_SIGNED_HALF)
    return SkyCubemap(np.clip(faces[..., :3] * 255 / 2, 0, 255).astype(np.uint8), abi.CUBEMAP_UNSIGNED_BYTE)


def make_params(scene, width, height, spp, trace_depth, seed=1, jitter=True, slice_offset=0, slice_divider=1,
                spp_max=None, extrema=(0.0, 0.0), diagnostics_stride=4, focus=None, sky_type=None, noise_color=None, noise_texture_index=0, rng_policy=0):
    """SampleBatchJob parameter block with the benchmark defaults of SURVEY.md section 8(d)."""
    p = abi.SampleParams()
    p.size = abi.Float2(float(width), float(height))
    p.sliceOffset = slice_offset
    p.sliceDivider = slice_divider
    p.seed = seed
    p.view = make_view(scene, width, height, focus)
    p.environment = abi.Environment(abi.SKY_GRADIENT if sky_type is None else sky_type, abi.Float3(*scene.sky_bottom), abi.Float3(*scene.sky_top))
    p.sampleCountRange[0] = spp
    p.sampleCountRange[1] = spp if spp_max is None else spp_max
    p.traceDepth = trace_depth
    p.subPixelJitter = 1 if jitter else 0
    p.noiseColor = abi.NOISE_WHITE if noise_color is None else noise_color
    p.sampleCountWeightExtrema = abi.Float2(float(extrema[0]), float(extrema[1]))
    p.diagnosticsStride = diagnostics_stride
    p.noiseTextureIndex = noise_texture_index
    p.rngPolicy = rng_policy
    return p
