"""Edge-case scene variants (used by tools/gen_golden.py to render reference fixtures and by the CPU / GPU tests).
Each returns .pbrt text; names are the fixture suffixes."""
import os, re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cornell(w=64, h=64, spp=4):
    t = open(os.path.join(ROOT, "scenes", "cornell.pbrt")).read()
    t = re.sub(r'"integer xresolution" \[\d+\] "integer yresolution" \[\d+\]', '"integer xresolution" [%d] "integer yresolution" [%d]' % (w, h), t)
    return re.sub(r'"integer pixelsamples" \[\d+\]', '"integer pixelsamples" [%d]' % spp, t)


_OPEN = '''LookAt 0 2.2 -6  0 0.8 0  0 1 0
Camera "perspective" "float fov" [40]
Sampler "sobol" "integer pixelsamples" [4]
PixelFilter "box"
Integrator "path" "integer maxdepth" [5]
Film "image" "integer xresolution" [72] "integer yresolution" [48] "string filename" "e.pfm"
WorldBegin
%s
Material "matte" "rgb Kd" [.5 .5 .5]
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-4 0 -4  4 0 -4  4 0 4  -4 0 4]
Material "glass" "float index" [1.5]
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3 4 6 5 4 7 6 0 4 5 0 5 1 1 5 6 1 6 2 2 6 7 2 7 3 3 7 4 3 4 0]
  "point P" [-1.6 0.01 -.5  -.6 0.01 -.5  -.6 0.01 .5  -1.6 0.01 .5  -1.6 1 -.5  -.6 1 -.5  -.6 1 .5  -1.6 1 .5]
Material "mirror"
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [.4 0 1.5  2.4 0 .5  2.4 1.6 .5  .4 1.6 1.5]
Material "plastic" "rgb Kd" [.7 .3 .2] "rgb Ks" [.3 .3 .3] "float roughness" [.1]
Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-.3 0 -1  .7 0 -1.2  .2 1.1 -1.1]
WorldEnd
'''


def scene(name):
    if name == "infinite":      # constant InfiniteAreaLight: escaped-ray emission, light sampling + MIS, two lights -> spatial strategy
        return _OPEN % ('LightSource "infinite" "rgb L" [.5 .6 .8]\nLightSource "point" "point from" [3 4 -2] "rgb I" [20 18 15]')
    if name == "infinite_only":  # a single light: CreateLightSampleDistribution substitutes uniform (lightdistrib.cpp:50)
        return _OPEN % 'LightSource "infinite" "rgb L" [.9 .8 .7]'
    if name == "spot":          # two spot lights, one under a transform (cone falloff, WorldToLight frame)
        return _OPEN % ('LightSource "spot" "point from" [2 4 -3] "point to" [-.5 0 0] "rgb I" [120 110 90] "float coneangle" [22] "float conedeltaangle" [7]\n'
                        'AttributeBegin\nRotate 20 0 1 0\nTranslate .3 0 0\n'
                        'LightSource "spot" "point from" [-3 3 -2] "point to" [0 .5 0] "rgb I" [40 60 90] "rgb scale" [.5 .5 .5]\nAttributeEnd')
    if name in ("envmap", "envmap_power"):   # infinite light with a radiance map (non-power-of-two: Lanczos resampling), rotated; + a point light
        t = _OPEN % ('AttributeBegin\nRotate -90 1 0 0\nRotate 30 0 0 1\nLightSource "infinite" "rgb L" [.8 .9 1] "string mapname" "%s"\nAttributeEnd\n'
                     'LightSource "point" "point from" [3 4 -2] "rgb I" [5 5 5]' % os.path.join(ROOT, "scenes", "envmap_40x20.pfm"))
        return t.replace('Integrator "path" "integer maxdepth" [5]', 'Integrator "path" "integer maxdepth" [5] "string lightsamplestrategy" "power"') if name == "envmap_power" else t
    if name == "instances":     # ObjectBegin / ObjectInstance (flattened to world space on the host) under rotations and non-uniform scales
        obj = ('ObjectBegin "thing"\nMaterial "plastic" "rgb Kd" [.2 .6 .3] "rgb Ks" [.3 .3 .3] "float roughness" [.1]\n'
               'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3 4 6 5 4 7 6 0 4 5 0 5 1 1 5 6 1 6 2 2 6 7 2 7 3 3 7 4 3 4 0]\n'
               '  "point P" [-.3 0 -.3  .3 0 -.3  .3 0 .3  -.3 0 .3  -.3 .6 -.3  .3 .6 -.3  .3 .6 .3  -.3 .6 .3]\nObjectEnd\n')
        inst = "".join('AttributeBegin\nTranslate %g 0 %g\nRotate %g 0 1 0\nScale %g %g %g\nObjectInstance "thing"\nAttributeEnd\n' % (x, z, a, sc, sc * 1.3, sc)
                       for x, z, a, sc in [(-1.5, 1, 20, 1), (0.2, 2, 75, .7), (1.6, .5, -40, 1.2), (-.4, -.5, 10, .5)])
        return _OPEN % ('LightSource "point" "point from" [3 4 -2] "rgb I" [30 28 25]\n'
                        'LightSource "distant" "point from" [-2 5 -3] "point to" [0 0 0] "rgb L" [1 1 1]\n' + obj + inst)
    if name == "spheres":       # Sphere primitives: glass sphere, clipped + transformed sphere, sphere area lights (one reversed, two-sided, clipped)
        sph = ('AttributeBegin\nTranslate -1.2 .7 .3\nMaterial "glass" "float index" [1.5]\nShape "sphere" "float radius" [.7]\nAttributeEnd\n'
               'AttributeBegin\nTranslate 1.2 .6 .8\nRotate 35 1 0 0\nScale 1 1.4 .8\n'
               'Material "plastic" "rgb Kd" [.7 .3 .2] "rgb Ks" [.3 .3 .3] "float roughness" [.1]\n'
               'Shape "sphere" "float radius" [.6] "float zmin" [-.3] "float zmax" [.45] "float phimax" [250]\nAttributeEnd\n'
               'AttributeBegin\nTranslate 0 2.6 -.5\nAreaLightSource "diffuse" "rgb L" [30 28 25]\nShape "sphere" "float radius" [.25]\nAttributeEnd\n'
               'AttributeBegin\nTranslate .2 .35 -1.2\nReverseOrientation\nAreaLightSource "diffuse" "rgb L" [3 5 8] "bool twosided" "true"\n'
               'Shape "sphere" "float radius" [.35] "float zmax" [.2]\nAttributeEnd\n')
        return _OPEN % sph
    if name == "dof":           # thin lens
        return _cornell().replace('Camera "perspective" "float fov" [39.3]', 'Camera "perspective" "float fov" [39.3] "float lensradius" [12] "float focaldistance" [1000]')
    if name == "crop":          # crop window + pixel bounds: partial tiles on every side, samples outside the bounds skipped
        t = _cornell(80, 56)
        t = t.replace('"string filename" "cornell.pfm"', '"string filename" "cornell.pfm" "float cropwindow" [.15 .8 .2 .9]')
        return t.replace('Integrator "path" "integer maxdepth" [5]', 'Integrator "path" "integer maxdepth" [5] "integer pixelbounds" [20 58 14 40]')
    if name == "clamp":         # Film maxsampleluminance
        return _cornell().replace('"string filename" "cornell.pfm"', '"string filename" "cornell.pfm" "float maxsampleluminance" [1.5]')
    if name == "empty":         # no geometry at all: every camera ray escapes to the environment
        return ('Camera "perspective" "float fov" [40]\nSampler "sobol" "integer pixelsamples" [2]\nPixelFilter "box"\nIntegrator "path"\n'
                'Film "image" "integer xresolution" [40] "integer yresolution" [24] "string filename" "e.pfm"\nWorldBegin\n'
                'LightSource "infinite" "rgb L" [.25 .5 1]\nWorldEnd\n')
    if name == "onetri":        # one triangle (single-leaf BVH), one degenerate triangle next to it, a point light
        return ('LookAt 0 0 -4  0 0 0  0 1 0\nCamera "perspective" "float fov" [40]\nSampler "sobol" "integer pixelsamples" [4]\nPixelFilter "box"\n'
                'Integrator "path" "integer maxdepth" [2]\nFilm "image" "integer xresolution" [48] "integer yresolution" [48] "string filename" "e.pfm"\nWorldBegin\n'
                'LightSource "point" "point from" [1 2 -3] "rgb I" [30 30 30]\nMaterial "matte" "rgb Kd" [.8 .6 .4]\n'
                'Shape "trianglemesh" "integer indices" [0 1 2 3 3 4] "point P" [-1 -1 0  1 -1 0  0 1 .5  .5 .5 0  .6 .6 0]\nWorldEnd\n')
    raise KeyError(name)


NAMES = ["infinite", "infinite_only", "envmap", "envmap_power", "spot", "instances", "spheres", "dof", "crop", "clamp", "empty", "onetri"]
