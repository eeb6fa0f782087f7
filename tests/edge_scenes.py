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


# ---- participating media: Integrator "volpath"
def _grid_density(nx, ny, nz):
    vals = []
    for z in range(nz):
        for y in range(ny):
            for x in range(nx):
                fx, fy, fz = (x + .5) / nx - .5, (y + .5) / ny - .5, (z + .5) / nz - .5
                r = (fx * fx + fy * fy + fz * fz) ** .5
                vals.append(max(0.0, 1.0 - 2.2 * r) * (1 + .5 * ((x * 7 + y * 3 + z * 5) % 4)))
    return " ".join("%.4g" % v for v in vals)


_SMOKE_BOX = ('Shape "trianglemesh" "integer indices" [0 1 2 0 2 3 4 6 5 4 7 6 0 4 5 0 5 1 1 5 6 1 6 2 2 6 7 2 7 3 3 7 4 3 4 0]\n'
              '  "point P" [%s]\n')


def vol_scene(name):
    volpath = lambda t: t.replace('Integrator "path" "integer maxdepth" [5]', 'Integrator "volpath" "integer maxdepth" [6]')
    lights = ('LightSource "point" "point from" [3 4 -2] "rgb I" [30 27 22]\n'
              'AttributeBegin\nAreaLightSource "diffuse" "rgb L" [12 12 10]\n'
              'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 3.5 -1  1 3.5 -1  1 3.5 1  -1 3.5 1]\nAttributeEnd\n')
    if name == "vol_fog":        # the camera and every surface sit in one homogeneous medium (chromatic sigma_t, forward-scattering HG)
        t = volpath(_OPEN % ('MediumInterface "" "fog"\n' + lights))
        return t.replace('LookAt', 'MakeNamedMedium "fog" "string type" "homogeneous" "rgb sigma_a" [.02 .03 .05] "rgb sigma_s" [.12 .1 .07] "float g" [.4]\n'
                                   'MediumInterface "" "fog"\nLookAt', 1)
    if name == "vol_smoke":      # a heterogeneous (grid) medium inside a box without a BSDF; vacuum elsewhere; an infinite light as well
        box = "-1 0.05 -1  1 0.05 -1  1 0.05 1  -1 0.05 1  -1 2.05 -1  1 2.05 -1  1 2.05 1  -1 2.05 1"
        med = ('AttributeBegin\nTranslate 0 .05 0\n'
               'MakeNamedMedium "smoke" "string type" "heterogeneous" "rgb sigma_a" [1 1 1] "rgb sigma_s" [5 5 5] "float g" [-.2]\n'
               '  "integer nx" [5] "integer ny" [6] "integer nz" [4] "point p0" [-1 0 -1] "point p1" [1 2 1] "float density" [%s]\nAttributeEnd\n'
               'AttributeBegin\nMediumInterface "smoke" ""\nMaterial ""\n%sAttributeEnd\n' % (_grid_density(5, 6, 4), _SMOKE_BOX % box))
        return volpath(_OPEN % ('LightSource "infinite" "rgb L" [.3 .35 .45]\n' + lights + med))
    if name == "vol_glass":      # a medium inside a glass box (an interface WITH a BSDF: refraction + the medium change), preset coefficients, fog outside
        t = volpath(_OPEN % ('MakeNamedMedium "milk" "string type" "homogeneous" "string preset" "Skimmilk" "float scale" [2]\n'
                             'MakeNamedMedium "haze" "string type" "homogeneous" "rgb sigma_a" [.01 .01 .01] "rgb sigma_s" [.04 .04 .05]\n'
                             'MediumInterface "" "haze"\n' + lights))
        t = t.replace('Material "glass" "float index" [1.5]', 'MediumInterface "milk" "haze"\nMaterial "glass" "float index" [1.3]')
        t = t.replace('Material "mirror"', 'MediumInterface "" "haze"\nMaterial "mirror"')
        return t.replace('LookAt', 'MakeNamedMedium "haze" "string type" "homogeneous" "rgb sigma_a" [.01 .01 .01] "rgb sigma_s" [.04 .04 .05]\nMediumInterface "" "haze"\nLookAt', 1)
    if name == "vol_alpha":      # alpha-masked meshes (no textured MATERIAL) in fog: VisibilityTester::Tr / Scene::IntersectTr go through Intersect, so
        t = tex_scene("tex_alpha")   # alpha masks count and shadow-alpha masks do not; + a grid medium in a BSDF-less box crossed by shadow rays (ratio tracking)
        t = t.replace('Integrator "path" "integer maxdepth" [4]', 'Integrator "volpath" "integer maxdepth" [4]')
        box = "-1.5 0.1 -2  1.5 0.1 -2  1.5 0.1 0  -1.5 0.1 0  -1.5 2.1 -2  1.5 2.1 -2  1.5 2.1 0  -1.5 2.1 0"
        med = ('AttributeBegin\nMakeNamedMedium "puff" "string type" "heterogeneous" "rgb sigma_a" [.5 .5 .5] "rgb sigma_s" [3 3 3] "float g" [.1]\n'
               '  "integer nx" [4] "integer ny" [3] "integer nz" [5] "point p0" [-1.5 .1 -2] "point p1" [1.5 2.1 0] "float density" [%s]\nAttributeEnd\n'
               'AttributeBegin\nMediumInterface "puff" "fog"\nMaterial ""\n%sAttributeEnd\n' % (_grid_density(4, 3, 5), _SMOKE_BOX % box))
        t = t.replace('WorldBegin\n', 'WorldBegin\nMediumInterface "" "fog"\n' + med, 1)
        return t.replace('LookAt', 'MakeNamedMedium "fog" "string type" "homogeneous" "rgb sigma_a" [.01 .015 .02] "rgb sigma_s" [.06 .05 .04] "float g" [.2]\n'
                                   'MediumInterface "" "fog"\nLookAt', 1)
    if name == "vol_alpha_fog":  # alpha masks under volpath with HOMOGENEOUS media only: the wavefront form walks the shadow / MIS rays with the closest-hit kernel, which
        t = tex_scene("tex_alpha")   # evaluates alphaMask (not shadowAlphaMask) exactly where VisibilityTester::Tr's Intersect does; + a dense homogeneous volume behind a BSDF-less box
        t = t.replace('Integrator "path" "integer maxdepth" [4]', 'Integrator "volpath" "integer maxdepth" [4]')
        box = "-1.5 0.1 -2  1.5 0.1 -2  1.5 0.1 0  -1.5 0.1 0  -1.5 2.1 -2  1.5 2.1 -2  1.5 2.1 0  -1.5 2.1 0"
        med = ('MakeNamedMedium "puff" "string type" "homogeneous" "rgb sigma_a" [.2 .2 .25] "rgb sigma_s" [.9 .8 .7] "float g" [.1]\n'
               'AttributeBegin\nMediumInterface "puff" "fog"\nMaterial ""\n%sAttributeEnd\n' % (_SMOKE_BOX % box))
        t = t.replace('WorldBegin\n', 'WorldBegin\nMediumInterface "" "fog"\n' + med, 1)
        return t.replace('LookAt', 'MakeNamedMedium "fog" "string type" "homogeneous" "rgb sigma_a" [.01 .015 .02] "rgb sigma_s" [.06 .05 .04] "float g" [.2]\n'
                                   'MediumInterface "" "fog"\nLookAt', 1)
    if name == "vol_inst":       # two-level instancing under volpath: instanced objects (one carrying its own inside medium behind a BSDF-less boundary, one of glass)
        # in fog, with rotations and a mirroring scale -- the per-lane transmittance / MIS rays of k_shade_vol enter and leave instances
        obj = ('ObjectBegin "cloud"\nMediumInterface "dense" "fog"\nMaterial ""\n'
               'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3 4 6 5 4 7 6 0 4 5 0 5 1 1 5 6 1 6 2 2 6 7 2 7 3 3 7 4 3 4 0]\n'
               '  "point P" [-.4 0 -.4  .4 0 -.4  .4 0 .4  -.4 0 .4  -.4 .8 -.4  .4 .8 -.4  .4 .8 .4  -.4 .8 .4]\nObjectEnd\n'
               'ObjectBegin "thing"\nMediumInterface "" "fog"\nMaterial "plastic" "rgb Kd" [.2 .6 .3] "rgb Ks" [.3 .3 .3] "float roughness" [.1]\n'
               'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3 4 6 5 4 7 6 0 4 5 0 5 1 1 5 6 1 6 2 2 6 7 2 7 3 3 7 4 3 4 0]\n'
               '  "point P" [-.3 0 -.3  .3 0 -.3  .3 0 .3  -.3 0 .3  -.3 .6 -.3  .3 .6 -.3  .3 .6 .3  -.3 .6 .3]\n'
               'AttributeBegin\nTranslate 0 1 0\nMaterial "glass" "float index" [1.4]\nShape "sphere" "float radius" [.3]\nAttributeEnd\nObjectEnd\n')
        inst = "".join('AttributeBegin\nTranslate %g %g %g\nRotate %g 0 1 0\nScale %g %g %g\nObjectInstance "%s"\nAttributeEnd\n' % a for a in
                       [(-1.5, .05, 1, 20, 1, 1, 1, "thing"), (1.6, .05, .5, -40, -1.1, 1.2, 1, "thing"), (.2, .06, 1.6, 30, 1, 1.2, 1, "cloud"), (-.6, .06, -.6, 10, .8, .8, 1.5, "cloud")])
        t = volpath(_OPEN % ('MakeNamedMedium "dense" "string type" "homogeneous" "rgb sigma_a" [.3 .2 .1] "rgb sigma_s" [2.5 2.8 3] "float g" [-.3]\n'
                             'MakeNamedMedium "fog" "string type" "homogeneous" "rgb sigma_a" [.02 .03 .05] "rgb sigma_s" [.12 .1 .07] "float g" [.4]\n'
                             'MediumInterface "" "fog"\n' + lights + obj + inst))
        return t.replace('LookAt', 'MakeNamedMedium "fog" "string type" "homogeneous" "rgb sigma_a" [.02 .03 .05] "rgb sigma_s" [.12 .1 .07] "float g" [.4]\n'
                                   'MediumInterface "" "fog"\nLookAt', 1)
    if name == "vol_none":       # volpath on a scene without any medium: the integrator's own differences from "path" (unconditional light sample, Intersect-based visibility)
        return volpath(_OPEN % lights)
    raise KeyError(name)


VOL_NAMES = ["vol_fog", "vol_smoke", "vol_glass", "vol_none", "vol_alpha", "vol_inst", "vol_alpha_fog"]


# ---- subsurface scattering: the BSSRDF branch of path / volpath
def sss_scene(name):
    lights = ('LightSource "point" "point from" [3 4 -2] "rgb I" [30 27 22]\n'
              'AttributeBegin\nAreaLightSource "diffuse" "rgb L" [12 12 10]\n'
              'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 3.5 -1  1 3.5 -1  1 3.5 1  -1 3.5 1]\nAttributeEnd\n')
    t = _OPEN % lights
    if name == "sss_named":    # measured coefficients by name (g forced to 0), smooth boundary, Integrator "path"
        return t.replace('Material "glass" "float index" [1.5]', 'Material "subsurface" "string name" "Skin1" "float scale" [12]')
    if name == "sss_coeff":    # explicit coefficients, g, eta, rough boundary (microfacet lobes), a second object of the same KIND but another material object
        t = t.replace('Material "glass" "float index" [1.5]',
                      'Material "subsurface" "rgb sigma_a" [.05 .2 .4] "rgb sigma_s" [6 5 3] "float g" [.3] "float eta" [1.5] "float uroughness" [.2] "float vroughness" [.1] "float scale" [3]')
        return t.replace('Material "plastic" "rgb Kd" [.7 .3 .2] "rgb Ks" [.3 .3 .3] "float roughness" [.1]',
                         'Material "subsurface" "rgb sigma_a" [.05 .2 .4] "rgb sigma_s" [6 5 3] "float g" [.3] "float eta" [1.5] "float uroughness" [.2] "float vroughness" [.1] "float scale" [3]')
    if name == "sss_kd":       # KdSubsurfaceMaterial: textured diffuse reflectance inverted to coefficients per hit (SubsurfaceFromDiffuse), under volpath in haze
        t = t.replace('Integrator "path" "integer maxdepth" [5]', 'Integrator "volpath" "integer maxdepth" [6]')
        t = t.replace('Material "glass" "float index" [1.5]',
                      'Texture "chk" "spectrum" "checkerboard" "float uscale" [3] "float vscale" [3] "rgb tex1" [.8 .3 .2] "rgb tex2" [.2 .5 .8]\n'
                      'Material "kdsubsurface" "texture Kd" "chk" "rgb mfp" [.4 .3 .2] "float eta" [1.4] "float scale" [.5]')
        return t.replace('LookAt', 'MakeNamedMedium "haze" "string type" "homogeneous" "rgb sigma_a" [.01 .01 .01] "rgb sigma_s" [.04 .04 .05]\nMediumInterface "" "haze"\nLookAt', 1)
    if name == "sss_vol_iface":   # sss_kd + a bank of denser fog behind a BSDF-less box that cuts through the subsurface object: the vertices' shadow / MIS rays are WALKED through the
        # interface, the probe chains cross it (its hits are not on the material object: not counted), the entry vertices sit in either medium
        fog = ('MakeNamedMedium "fog2" "string type" "homogeneous" "rgb sigma_a" [.05 .05 .06] "rgb sigma_s" [.3 .3 .35] "float g" [.3]\n'
               'AttributeBegin\nMediumInterface "fog2" "haze"\nMaterial ""\n'
               'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3 4 6 5 4 7 6 0 4 5 0 5 1 1 5 6 1 6 2 2 6 7 2 7 3 3 7 4 3 4 0]\n'
               '  "point P" [-2 .005 -1.2  -.2 .005 -1.2  -.2 .005 .1  -2 .005 .1  -2 1.3 -1.2  -.2 1.3 -1.2  -.2 1.3 .1  -2 1.3 .1]\nAttributeEnd\n')
        return sss_scene("sss_kd").replace("WorldEnd", fog + "WorldEnd")
    if name == "sss_vol_smoke":   # sss_kd + a HETEROGENEOUS medium behind a BSDF-less box around the subsurface object: every transmittance query of the subsurface vertex and of the entry
        # vertex draws ratio-tracking dimensions -- both vertices are shaded in two stages (split form), the probe chain in between
        smoke = ('AttributeBegin\nTranslate -1.1 .005 0\n'
                 'MakeNamedMedium "smoke" "string type" "heterogeneous" "rgb sigma_a" [.3 .3 .3] "rgb sigma_s" [1.5 1.5 1.5] "float g" [.2]\n'
                 '  "integer nx" [4] "integer ny" [3] "integer nz" [4] "point p0" [-1 0 -1] "point p1" [1 1.5 1] "float density" [%s]\nAttributeEnd\n'
                 'AttributeBegin\nMediumInterface "smoke" "haze"\nMaterial ""\n'
                 'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3 4 6 5 4 7 6 0 4 5 0 5 1 1 5 6 1 6 2 2 6 7 2 7 3 3 7 4 3 4 0]\n'
                 '  "point P" [-2.1 .005 -1  -.1 .005 -1  -.1 .005 1  -2.1 .005 1  -2.1 1.505 -1  -.1 1.505 -1  -.1 1.505 1  -2.1 1.505 1]\nAttributeEnd\n') % _grid_density(4, 3, 4)
        return sss_scene("sss_kd").replace("WorldEnd", smoke + "WorldEnd")
    if name == "sss_inst":     # a subsurface object INSTANTIATED twice (two-level instancing): probe-ray chains through TransformedPrimitives, under "path"
        obj = ('ObjectBegin "blob"\nMaterial "subsurface" "string name" "Ketchup" "float scale" [6] "float eta" [1.35]\n' + _bulge() + 'ObjectEnd\n')
        inst = "".join('AttributeBegin\nTranslate %g %g %g\nRotate %g 0 1 0\nScale %g %g %g\nObjectInstance "blob"\nAttributeEnd\n' % a for a in
                       [(-2.2, 0, 1.2, 25, 1, 1, 1), (1.9, 0, .8, -50, .8, 1.2, -.9)])
        return t.replace('WorldBegin\n', 'WorldBegin\n' + obj + inst, 1)
    raise KeyError(name)


SSS_NAMES = ["sss_named", "sss_coeff", "sss_kd", "sss_inst", "sss_vol_iface", "sss_vol_smoke"]


# ---- the samplers that draw from one PCG32 stream per tile (ABI v11): RandomSampler, StratifiedSampler, ZeroTwoSequenceSampler
def sampler_scene(name):
    def with_sampler(text, line):
        out, n = re.subn(r'Sampler "[a-z0-9]+"[^\n]*', line, text, count=1)
        assert n == 1
        return out
    if name == "sampler_random":       # every value from the tile's stream; 72 x 48: partial tiles on the right
        return with_sampler(scene("infinite"), 'Sampler "random" "integer pixelsamples" [3]')
    if name == "sampler_stratified":   # 2 x 3 strata, jittered, 4 precomputed dimensions; dof: the lens sample is 2D dimension 1
        return with_sampler(scene("dof"), 'Sampler "stratified" "integer xsamples" [2] "integer ysamples" [3]')
    if name == "sampler_strat_d1":     # no jitter, ONE precomputed dimension: the lens sample and everything after it come from the stream (Point2f(a(), b()) order)
        return with_sampler(scene("spot"), 'Sampler "stratified" "integer xsamples" [2] "integer ysamples" [2] "bool jitter" ["false"] "integer dimensions" [1]')
    if name == "sampler_02sequence":   # crop window + pixel bounds: StartPixel runs for the pixels outside the bounds too (integrator.cpp:262-273)
        return with_sampler(scene("crop"), 'Sampler "02sequence" "integer pixelsamples" [4]')
    if name == "sampler_lowdisc_vol":  # "lowdiscrepancy" = 02sequence, 3 -> 4 samples; volpath in fog: data-dependent numbers of draws per path
        return with_sampler(scene("vol_fog"), 'Sampler "lowdiscrepancy" "integer pixelsamples" [3] "integer dimensions" [2]')
    if name == "sampler_maxmin":       # MaxMinDistSampler (ABI v12): 6 -> 8 samples, 3 precomputed dimensions; depth of field: the lens sample is 2D dimension 1 (Sobol2D), the film
        # sample 2D dimension 0 (the CMaxMinDist matrix).  Only through the reference-side binding: the matrix is the reference's (STUB_ONLY_SAMPLER_NAMES)
        return with_sampler(scene("dof"), 'Sampler "maxmindist" "integer pixelsamples" [6] "integer dimensions" [3]')
    raise KeyError(name)


SAMPLER_NAMES = ["sampler_random", "sampler_stratified", "sampler_strat_d1", "sampler_02sequence", "sampler_lowdisc_vol"]
STUB_ONLY_SAMPLER_NAMES = ["sampler_maxmin"]   # this repository's own host has no CMaxMinDist table (it renders such scenes with sobol + a warning)


def scene(name):
    if name.startswith("sampler_"):
        return sampler_scene(name)
    if name.startswith("tex_"):
        return tex_scene(name)
    if name.startswith("vol_"):
        return vol_scene(name)
    if name.startswith("sss_"):
        return sss_scene(name)
    if name == "infinite":      # constant InfiniteAreaLight: escaped-ray emission, light sampling + MIS, two lights -> spatial strategy
        return _OPEN % ('LightSource "infinite" "rgb L" [.5 .6 .8]\nLightSource "point" "point from" [3 4 -2] "rgb I" [20 18 15]')
    if name == "infinite_xf":   # a CONSTANT infinite light under a rotation + mirroring scale: LightToWorld decides where a sample (u,v) points (infinite.cpp:116-118)
        return _OPEN % ('AttributeBegin\nRotate 50 1 .3 0\nScale -1 1 1\nLightSource "infinite" "rgb L" [.5 .6 .8]\nAttributeEnd\n'
                        'LightSource "point" "point from" [3 4 -2] "rgb I" [20 18 15]')
    if name == "infinite_only":  # a single light: CreateLightSampleDistribution substitutes uniform (lightdistrib.cpp:50)
        return _OPEN % 'LightSource "infinite" "rgb L" [.9 .8 .7]'
    if name == "spot":          # two spot lights, one under a transform (cone falloff, WorldToLight frame)
        return _OPEN % ('LightSource "spot" "point from" [2 4 -3] "point to" [-.5 0 0] "rgb I" [120 110 90] "float coneangle" [22] "float conedeltaangle" [7]\n'
                        'AttributeBegin\nRotate 20 0 1 0\nTranslate .3 0 0\n'
                        'LightSource "spot" "point from" [-3 3 -2] "point to" [0 .5 0] "rgb I" [40 60 90] "rgb scale" [.5 .5 .5]\nAttributeEnd')
    if name == "envmap_png":    # the radiance map read from an 8-bit PNG (host/imageread.cpp instead of the PFM reader); same device path as "envmap"
        return scene("envmap").replace(os.path.join(ROOT, "scenes", "envmap_40x20.pfm"), os.path.join(ROOT, "scenes", "textures", "color_23x17.png"))
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
    if name == "instances2":    # instancing stress: an object with a triangle mesh (shading normals, textured material, alpha mask) and a sphere,
        # instantiated under rotations, a non-uniform scale and a MIRRORING scale; a one-triangle object (no BVH in the reference)
        obj = ('Texture "chk" "color" "checkerboard" "float uscale" [4] "float vscale" [4] "rgb tex1" [.8 .2 .2] "rgb tex2" [.9 .9 .8]\n'
               'Texture "msk" "float" "checkerboard" "string aamode" "none" "float uscale" [3] "float vscale" [3] "float tex1" [1] "float tex2" [0]\n'
               'ObjectBegin "thing"\nMaterial "plastic" "texture Kd" "chk" "rgb Ks" [.3 .3 .3] "float roughness" [.1]\n' + _bulge().replace('"float uv"', '"texture alpha" "msk" "float uv"') +
               'AttributeBegin\nTranslate 1.4 .9 0\nMaterial "glass" "float index" [1.5]\nShape "sphere" "float radius" [.35]\nAttributeEnd\nObjectEnd\n'
               'ObjectBegin "one"\nMaterial "matte" "rgb Kd" [.2 .3 .8]\nShape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0  .8 0 0  .4 .9 0]\nObjectEnd\n')
        inst = "".join('AttributeBegin\nTranslate %g %g %g\nRotate %g 0 1 0\nScale %g %g %g\nObjectInstance "%s"\nAttributeEnd\n' % a for a in
                       [(-2.6, 0, .5, 25, 1, 1, 1, "thing"), (.2, .1, 1.6, -60, .7, 1.3, .7, "thing"), (2.2, 0, -.2, 140, -.8, .8, .8, "thing"),
                        (-.6, .05, -1.2, 10, 1, 1, 1, "one"), (1.1, .05, -1.0, -35, 1.5, .7, 1, "one")])
        return _OPEN % ('LightSource "point" "point from" [3 4 -2] "rgb I" [30 28 25]\n'
                        'LightSource "distant" "point from" [-2 5 -3] "point to" [0 0 0] "rgb L" [1 1 1]\n' + obj + inst)
    if name == "nurbs":         # Shape "nurbs": a rational bicubic patch (Pw, non-uniform knots, u0/u1 sub-range) and a biquadratic one (P), diced by the reference itself
        import math
        pw = []
        for j in range(4):
            for i in range(5):
                wgt = 1 + .6 * ((i + j) % 2)
                x, y, z = i * .5 - 1, .45 * math.sin(1.1 * i + .7 * j), j * .5 - .75
                pw += [x * wgt, y * wgt, z * wgt, wgt]
        p2 = [c for j in range(3) for i in range(3) for c in (i * .6, .3 * ((i * j) % 2), j * .6)]
        fmt = lambda xs: " ".join("%.6g" % x for x in xs)
        return _OPEN % ('LightSource "point" "point from" [3 4 -2] "rgb I" [30 28 25]\n'
                        'Texture "chk" "color" "checkerboard" "float uscale" [8] "float vscale" [8] "rgb tex1" [.8 .3 .2] "rgb tex2" [.9 .9 .8]\n'
                        'AttributeBegin\nTranslate -.4 .7 -.6\nMaterial "plastic" "texture Kd" "chk" "rgb Ks" [.3 .3 .3] "float roughness" [.08]\n'
                        'Shape "nurbs" "integer nu" [5] "integer nv" [4] "integer uorder" [4] "integer vorder" [4] "float uknots" [0 0 0 0 .4 1 1 1 1] "float vknots" [0 0 0 0 1 1 1 1] '
                        '"float u0" [.05] "float u1" [.95] "float Pw" [%s]\nAttributeEnd\n'
                        'AttributeBegin\nTranslate 1.1 .3 -1.4\nRotate 20 0 1 0\nMaterial "matte" "rgb Kd" [.3 .4 .8]\n'
                        'Shape "nurbs" "integer nu" [3] "integer nv" [3] "integer uorder" [3] "integer vorder" [3] "float uknots" [0 0 0 1 1 1] "float vknots" [0 0 0 1 1 1] "point P" [%s]\nAttributeEnd\n'
                        % (fmt(pw), fmt(p2)))
    if name == "heightfield":   # Shape "heightfield": the reference's own tessellation into a uv-mapped triangle mesh (heightfield.cpp), under a transform
        import math
        n = 9
        z = " ".join("%.5g" % (.25 * math.sin(1.3 * x) * math.cos(.9 * y) + .3) for y in range(n) for x in range(n))
        return _OPEN % ('LightSource "point" "point from" [3 4 -2] "rgb I" [30 28 25]\n'
                        'Texture "chk" "color" "checkerboard" "float uscale" [6] "float vscale" [6] "rgb tex1" [.8 .3 .2] "rgb tex2" [.9 .9 .8]\n'
                        'AttributeBegin\nTranslate -1.2 .05 -1.5\nRotate -90 1 0 0\nScale 2.4 2.2 1\nMaterial "matte" "texture Kd" "chk"\n'
                        'Shape "heightfield" "integer nu" [%d] "integer nv" [%d] "float Pz" [%s]\nAttributeEnd\n' % (n, n, z))
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
    if name == "c5_crop":       # configs[4]'s regime end to end: 3840 x 2160 film (m = 12), 512 spp => 33-bit Sobol' indices; a 24 x 20 pixel crop far from the origin
        return _cornell(3840, 2160, 512).replace('"string filename" "cornell.pfm"', '"string filename" "cornell.pfm" "float cropwindow" [.55 .55625 .45 .45926]')
    if name == "clamp":         # Film maxsampleluminance
        return _cornell().replace('"string filename" "cornell.pfm"', '"string filename" "cornell.pfm" "float maxsampleluminance" [1.5]')
    if name == "spectra":       # "blackbody" and "spectrum" parameters (inline samples and SPD files): parser.cpp:662-690 -> RGB through the CIE tables
        spd = os.path.join(ROOT, "scenes", "spds")
        t = _OPEN % ('LightSource "point" "point from" [3 4 -2] "blackbody I" [3200 18]\n'
                     'LightSource "distant" "point from" [-2 5 -3] "point to" [0 0 0] "blackbody L" [6500 .8]')
        t = t.replace('Material "matte" "rgb Kd" [.5 .5 .5]', 'Material "matte" "spectrum Kd" [400 .2 500 .35 600 .7 700 .75 650 .72]')   # unsorted on purpose
        return t.replace('Material "mirror"', 'Material "metal" "spectrum eta" "%s/testmetal.eta.spd" "spectrum k" "%s/testmetal.k.spd" "float roughness" [.08]' % (spd, spd))
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


NAMES = ["infinite", "infinite_only", "envmap", "envmap_power", "spot", "instances", "spheres", "dof", "crop", "clamp", "empty", "onetri", "spectra"]

# ---- textured variants: image / procedural textures, mappings, bump maps, alpha masks
C5_NAMES = ["c5_crop"]
TEX = os.path.join(ROOT, "scenes", "textures")
_TEXHEAD = '''LookAt 0 2.4 -6  0 0.7 0  0 1 0
Camera "perspective" "float fov" [38]
Sampler "sobol" "integer pixelsamples" [4]
PixelFilter "box"
Integrator "path" "integer maxdepth" [4]
Film "image" "integer xresolution" [72] "integer yresolution" [48] "string filename" "e.pfm"
WorldBegin
LightSource "point" "point from" [3 5 -3] "rgb I" [60 56 50]
LightSource "distant" "point from" [-2 5 -3] "point to" [0 0 0] "rgb L" [1.5 1.5 1.6]
'''
# ground quad (uv 0..4), a tilted panel (uv 0..1), a smooth-normal "bulge" mesh (3x3 grid with per-vertex normals and uv), a back wall
_GROUND = 'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-4 0 -4  4 0 -4  4 0 4  -4 0 4] "float uv" [0 0 4 0 4 4 0 4]\n'
_PANEL = 'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-2.4 0.1 .6  -.6 0.1 1.2  -.6 1.7 1.2  -2.4 1.7 .6] "float uv" [0 0 1 0 1 1 0 1]\n'
_WALL = 'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-4 0 3  4 0 3  4 3 3  -4 3 3] "float uv" [0 0 2 0 2 1 0 1]\n'


def _bulge():
    import math
    P, N, UV, I = [], [], [], []
    n = 5
    for j in range(n):
        for i in range(n):
            u, v = i / (n - 1), j / (n - 1)
            x, z = .5 + 1.8 * u, -.6 + 1.4 * v
            hgt = .25 + .35 * math.sin(math.pi * u) * math.sin(math.pi * v)
            dhx = .35 * math.pi * math.cos(math.pi * u) * math.sin(math.pi * v) / 1.8
            dhz = .35 * math.pi * math.sin(math.pi * u) * math.cos(math.pi * v) / 1.4
            l = math.sqrt(dhx * dhx + 1 + dhz * dhz)
            P += [x, hgt, z]; N += [-dhx / l, 1 / l, -dhz / l]; UV += [u, v]
    for j in range(n - 1):
        for i in range(n - 1):
            a = j * n + i
            I += [a, a + n, a + 1, a + 1, a + n, a + n + 1]
    f = lambda xs: " ".join("%.6g" % x for x in xs)
    return 'Shape "trianglemesh" "integer indices" [%s] "point P" [%s] "normal N" [%s] "float uv" [%s]\n' % (" ".join(map(str, I)), f(P), f(N), f(UV))


def tex_scene(name):
    T = lambda f: os.path.join(TEX, f)
    w = _TEXHEAD
    if name == "tex_imagemap":      # EWA (default) lookups of a non-power-of-two PNG (gamma), a TGA with clamp wrap + trilinear, an HDR PFM with scale; float imagemap as roughness
        w += ('Texture "col" "color" "imagemap" "string filename" "%s" "float uscale" [2] "float vscale" [2]\n' % T("color_23x17.png") +
              'Texture "noise" "color" "imagemap" "string filename" "%s" "bool trilinear" ["true"] "string wrap" "clamp" "float udelta" [-.2]\n' % T("noise_16x8.tga") +
              'Texture "hdr" "color" "imagemap" "string filename" "%s" "float scale" [.6] "string wrap" "black" "float maxanisotropy" [4]\n' % T("hdr_12x10.pfm") +
              'Texture "rough" "float" "imagemap" "string filename" "%s" "float scale" [.4] "bool gamma" ["false"]\n' % T("height_32.png") +
              'Texture "pal" "color" "imagemap" "string filename" "%s"\n' % T("palette_8.png") +
              'Material "matte" "texture Kd" "col"\n' + _GROUND +
              'Material "plastic" "texture Kd" "noise" "rgb Ks" [.4 .4 .4] "texture roughness" "rough"\n' + _PANEL +
              'Material "matte" "texture Kd" "hdr" "float sigma" [20]\n' + _bulge() +
              'Material "matte" "texture Kd" "pal"\n' + _WALL)
    elif name == "tex_spheres":     # textured / bump-mapped materials ON SPHERES (uv = (phi / phiMax, (theta - thetaMin) / dTheta), dndu / dndv from the fundamental
        # forms): a full sphere, a clipped one under a non-uniform scale + mirroring, one inside an instantiated object
        w += ('Texture "col" "color" "imagemap" "string filename" "%s" "float uscale" [3] "float vscale" [2]\n' % T("color_23x17.png") +
              'Texture "ch" "color" "checkerboard" "float uscale" [8] "float vscale" [6] "rgb tex1" [.8 .1 .1] "rgb tex2" [.9 .9 .8]\n'
              'Texture "hgt" "float" "imagemap" "string filename" "%s" "float scale" [.08] "bool gamma" ["false"] "float uscale" [2]\n' % T("height_32.png") +
              'Texture "wr" "float" "wrinkled" "integer octaves" [3]\nTexture "wrs" "float" "scale" "texture tex1" "wr" "float tex2" [.04]\n'
              'Texture "rough" "float" "checkerboard" "string aamode" "none" "float uscale" [5] "float vscale" [5] "float tex1" [.02] "float tex2" [.3]\n'
              'Material "matte" "rgb Kd" [.6 .6 .6]\n' + _GROUND +
              'AttributeBegin\nTranslate -1.6 .8 .4\nRotate 30 0 1 0\nMaterial "plastic" "texture Kd" "col" "rgb Ks" [.3 .3 .3] "texture roughness" "rough" "texture bumpmap" "hgt"\n'
              'Shape "sphere" "float radius" [.8]\nAttributeEnd\n'
              'AttributeBegin\nTranslate .6 .7 -.3\nRotate -70 1 0 0\nScale -1 1.3 1\nMaterial "matte" "texture Kd" "ch" "texture bumpmap" "wrs"\n'
              'Shape "sphere" "float radius" [.7] "float zmin" [-.5] "float zmax" [.6] "float phimax" [300]\nAttributeEnd\n'
              'ObjectBegin "ball"\nMaterial "uber" "texture Kd" "ch" "rgb Ks" [.2 .2 .2] "texture roughness" "rough"\nShape "sphere" "float radius" [.45]\nObjectEnd\n'
              'AttributeBegin\nTranslate 2.2 .5 .8\nRotate 40 0 0 1\nScale 1 .8 1.2\nObjectInstance "ball"\nAttributeEnd\n'
              'AttributeBegin\nTranslate -.2 .45 1.9\nObjectInstance "ball"\nAttributeEnd\n' +
              'Material "matte" "rgb Kd" [.5 .55 .6]\n' + _WALL)
    elif name == "tex_procedural":  # checkerboard 2D (closed form + none) and 3D, dots, uv, bilerp, mix and scale of textures
        w += ('Texture "ch" "color" "checkerboard" "float uscale" [6] "float vscale" [6] "rgb tex1" [.8 .1 .1] "rgb tex2" [.9 .9 .8]\n'
              'Texture "chn" "color" "checkerboard" "string aamode" "none" "float uscale" [3] "float vscale" [5] "rgb tex1" [.1 .1 .7] "rgb tex2" [.8 .8 .2]\n'
              'TransformBegin\nScale .5 .5 .5\nRotate 30 0 1 0\nTexture "ch3" "color" "checkerboard" "integer dimension" [3] "rgb tex1" [.2 .7 .2] "rgb tex2" [.7 .7 .7]\nTransformEnd\n'
              'Texture "dots" "color" "dots" "float uscale" [5] "float vscale" [5] "rgb inside" [.9 .6 .1] "texture outside" "chn"\n'
              'Texture "uvt" "color" "uv" "float uscale" [2] "float vscale" [3] "float udelta" [.25]\n'
              'Texture "bil" "color" "bilerp" "rgb v00" [1 0 0] "rgb v01" [0 1 0] "rgb v10" [0 0 1] "rgb v11" [1 1 0]\n'
              'Texture "amt" "float" "checkerboard" "float uscale" [2] "float vscale" [2] "float tex1" [.2] "float tex2" [.9]\n'
              'Texture "mixd" "color" "mix" "texture tex1" "uvt" "texture tex2" "bil" "texture amount" "amt"\n'
              'Texture "scl" "color" "scale" "texture tex1" "mixd" "rgb tex2" [.9 .8 .7]\n'
              'Material "matte" "texture Kd" "ch"\n' + _GROUND +
              'Material "matte" "texture Kd" "dots"\n' + _PANEL +
              'Material "matte" "texture Kd" "ch3"\n' + _bulge() +
              'Material "matte" "texture Kd" "scl"\n' + _WALL)
    elif name == "tex_noise":       # Perlin noise family under a texture-space transform: fbm, wrinkled, marble, windy
        w += ('TransformBegin\nScale .4 .4 .4\nRotate 25 1 0 0\n'
              'Texture "fbm" "float" "fbm" "integer octaves" [5] "float roughness" [.6]\n'
              'Texture "wr" "color" "wrinkled" "integer octaves" [6]\n'
              'Texture "mar" "color" "marble" "float scale" [2.5] "float variation" [.3]\n'
              'Texture "wi" "float" "windy"\nTransformEnd\n'
              'Texture "fbmc" "color" "scale" "texture tex1" "wr" "rgb tex2" [.9 .7 .5]\n'
              'Material "matte" "texture Kd" "mar"\n' + _GROUND +
              'Material "matte" "texture Kd" "fbmc" "texture sigma" "fbm"\n' + _PANEL +
              'Material "plastic" "texture Kd" "wr" "rgb Ks" [.3 .3 .3] "texture roughness" "wi"\n' + _bulge() +
              'Material "matte" "texture Kd" "mar"\n' + _WALL)
    elif name == "tex_mappings":    # spherical, cylindrical and planar 2D mappings of an image and a checkerboard
        w += ('TransformBegin\nTranslate 0 1 0\nRotate 40 0 1 0\n'
              'Texture "sph" "color" "imagemap" "string filename" "%s" "string mapping" "spherical"\n' % T("color_23x17.png") +
              'Texture "cyl" "color" "checkerboard" "string mapping" "cylindrical" "rgb tex1" [.8 .2 .2] "rgb tex2" [.9 .9 .9]\nTransformEnd\n'
              'Texture "pln" "color" "imagemap" "string filename" "%s" "string mapping" "planar" "vector v1" [.5 0 .2] "vector v2" [0 .6 .1] "float udelta" [.1] "float vdelta" [.3]\n' % T("noise_16x8.tga") +
              'Material "matte" "texture Kd" "sph"\n' + _GROUND +
              'Material "matte" "texture Kd" "pln"\n' + _PANEL +
              'Material "matte" "texture Kd" "cyl"\n' + _bulge() +
              'Material "matte" "texture Kd" "pln"\n' + _WALL)
    elif name == "tex_bump":        # Material::Bump with an image height field and with fbm, on flat and on shading-normal geometry
        w += ('Texture "h" "float" "imagemap" "string filename" "%s" "float scale" [.08] "bool gamma" ["false"] "float uscale" [2] "float vscale" [2]\n' % T("height_32.png") +
              'TransformBegin\nScale .3 .3 .3\nTexture "f" "float" "fbm" "integer octaves" [4]\nTransformEnd\n'
              'Texture "fs" "float" "scale" "texture tex1" "f" "float tex2" [.05]\n'
              'Material "plastic" "rgb Kd" [.4 .5 .6] "rgb Ks" [.3 .3 .3] "float roughness" [.15] "texture bumpmap" "h"\n' + _GROUND +
              'Material "matte" "rgb Kd" [.7 .6 .4] "texture bumpmap" "fs"\n' + _PANEL +
              'Material "plastic" "rgb Kd" [.6 .3 .3] "rgb Ks" [.4 .4 .4] "float roughness" [.1] "texture bumpmap" "h"\n' + _bulge() +
              'Material "mirror" "rgb Kr" [.8 .8 .8] "texture bumpmap" "fs"\n' + _WALL)
    elif name == "tex_alpha":       # alpha masks: image mask (camera + shadow rays), shadow-only mask, constant "float alpha" 0
        w += ('Texture "m" "float" "imagemap" "string filename" "%s" "bool gamma" ["false"] "string wrap" "clamp"\n' % T("mask_16.png") +
              'Texture "stripes" "float" "checkerboard" "string aamode" "none" "float uscale" [8] "float vscale" [1] "float tex1" [0] "float tex2" [1]\n'
              'Material "matte" "rgb Kd" [.6 .6 .6]\n' + _GROUND +
              'Material "matte" "rgb Kd" [.8 .3 .2]\n' + _PANEL.replace('"float uv"', '"texture alpha" "m" "float uv"') +
              'Material "matte" "rgb Kd" [.2 .5 .8]\n' + _bulge().replace('"float uv"', '"texture shadowalpha" "stripes" "float uv"') +
              'Material "matte" "rgb Kd" [.5 .5 .5]\n' + _WALL.replace('"float uv"', '"float alpha" [0] "float uv"') +
              'Material "matte" "rgb Kd" [.3 .7 .3]\n'
              'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-4 0 3.5  4 0 3.5  4 3 3.5  -4 3 3.5]\n')
    elif name == "tex_materials":   # textured parameters of the other materials: uber (opacity map), glass (rough map), metal, substrate, translucent, mix (textured amount)
        w += ('Texture "col" "color" "imagemap" "string filename" "%s"\n' % T("color_23x17.png") +
              'Texture "op" "color" "checkerboard" "float uscale" [4] "float vscale" [4] "rgb tex1" [1 1 1] "rgb tex2" [.2 .2 .2]\n'
              'Texture "r" "float" "imagemap" "string filename" "%s" "float scale" [.3] "bool gamma" ["false"]\n' % T("height_32.png") +
              'Texture "amt" "color" "checkerboard" "float uscale" [3] "float vscale" [3] "rgb tex1" [.1 .1 .1] "rgb tex2" [.9 .9 .9]\n'
              'MakeNamedMaterial "a" "string type" "matte" "texture Kd" "col"\n'
              'MakeNamedMaterial "b" "string type" "metal" "texture roughness" "r"\n'
              'Material "substrate" "texture Kd" "col" "rgb Ks" [.2 .2 .2] "texture uroughness" "r" "float vroughness" [.05]\n' + _GROUND +
              'Material "uber" "texture Kd" "col" "rgb Ks" [.2 .2 .2] "rgb Kr" [.1 .1 .1] "texture opacity" "op" "texture roughness" "r"\n' + _PANEL +
              'Material "mix" "string namedmaterial1" "a" "string namedmaterial2" "b" "texture amount" "amt"\n' + _bulge() +
              'Material "translucent" "texture Kd" "col" "rgb Ks" [.2 .2 .2] "texture roughness" "r"\n' + _WALL +
              'Material "glass" "texture uroughness" "r" "float vroughness" [.02] "float index" [1.4]\n'
              'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-.4 0.1 -1.6  .6 0.1 -1.9  .6 1.0 -1.9  -.4 1.0 -1.6] "float uv" [0 0 1 0 1 1 0 1]\n')
    elif name == "tex_dof":         # thin-lens camera (the lens branch of the ray differentials), Halton sampler, trilinear + EWA image maps, a bump map
        w = w.replace('Camera "perspective" "float fov" [38]', 'Camera "perspective" "float fov" [38] "float lensradius" [.12] "float focaldistance" [6.2]')
        w = w.replace('Sampler "sobol" "integer pixelsamples" [4]', 'Sampler "halton" "integer pixelsamples" [5]')
        w += ('Texture "col" "color" "imagemap" "string filename" "%s" "float uscale" [3] "float vscale" [3]\n' % T("color_23x17.png") +
              'Texture "tri" "color" "imagemap" "string filename" "%s" "bool trilinear" ["true"]\n' % T("hdr_12x10.pfm") +
              'Texture "h" "float" "imagemap" "string filename" "%s" "float scale" [.05] "bool gamma" ["false"]\n' % T("height_32.png") +
              'Material "matte" "texture Kd" "col"\n' + _GROUND +
              'Material "plastic" "texture Kd" "tri" "rgb Ks" [.3 .3 .3] "float roughness" [.1] "texture bumpmap" "h"\n' + _PANEL +
              'Material "matte" "texture Kd" "col" "texture bumpmap" "h"\n' + _bulge() +
              'Material "matte" "texture Kd" "tri"\n' + _WALL)
    else:
        raise KeyError(name)
    return w + "WorldEnd\n"


TEX_NAMES = ["tex_imagemap", "tex_procedural", "tex_noise", "tex_mappings", "tex_bump", "tex_alpha", "tex_materials", "tex_spheres"]
# pinned for the oracle only so far (the device tests of these run from the round in which they were first measured on a GPU)
TEX_ORACLE_ONLY = ["tex_dof", "envmap_png", "heightfield", "infinite_xf", "nurbs"]
INSTANCE_NAMES = ["instances", "instances2"]   # object instancing: flattened by default, two-level with PBRT_AMD_INSTANCING=1 (oracle)


# ---- the reference's analytic "furnace" scenes (src/tests/analytic_scenes.cpp:71-203, rendered there by :268-330 and checked by
# CheckSceneAverage :55-68: mean of all channels 1.0 +- 0.02): the camera sits at the centre of a unit sphere seen from inside
FURNACE_NAMES = ["furnace_point", "furnace_4points", "furnace_area", "furnace_uber"]


def furnace_scene(name, sampler="sobol", integrator="path"):
    world = {
        "furnace_point": 'LightSource "point" "rgb I" [3.14159265 3.14159265 3.14159265]\nMaterial "matte" "rgb Kd" [.5 .5 .5] "float sigma" [0]\n',
        "furnace_4points": 'LightSource "point" "rgb I" [.785398163 .785398163 .785398163]\n' * 4 + 'Material "matte" "rgb Kd" [.5 .5 .5] "float sigma" [0]\n',
        "furnace_area": 'Material "matte" "rgb Kd" [.5 .5 .5] "float sigma" [0]\nAreaLightSource "diffuse" "rgb L" [.5 .5 .5]\n',
        "furnace_uber": ('LightSource "point" "rgb I" [9.42477796 9.42477796 9.42477796]\n'
                         'Material "uber" "rgb Kd" [.25 .25 .25] "rgb Ks" [0 0 0] "rgb Kr" [.5 .5 .5] "rgb Kt" [0 0 0] "float roughness" [0] '
                         '"rgb opacity" [1 1 1] "float eta" [1] "bool remaproughness" "false"\n'),
    }[name]
    return ('Camera "perspective" "float fov" [45]\nSampler "%s" "integer pixelsamples" [256]\nPixelFilter "box"\n'
            'Integrator "%s" "integer maxdepth" [8]\nFilm "image" "integer xresolution" [10] "integer yresolution" [10] "string filename" "f.pfm"\n'
            'WorldBegin\n%sReverseOrientation\nShape "sphere" "float radius" [1]\nWorldEnd\n' % (sampler, integrator, world))


# ---- light-sampling known-answer records (tests/golden/light_vectors.npz, from oracle/ref_build/ref_probe.cpp): one light per 24 records
def light_kat_scene(recs):
    """the scene whose light i is the light of records [24 i, 24 i + 24): triangle / sphere DiffuseAreaLights, point and spot lights"""
    def fl(v):
        return " ".join("%.9g" % x for x in v)
    out = ['Film "image" "integer xresolution" [8] "integer yresolution" [8] "string filename" "l.pfm"\nSampler "sobol" "integer pixelsamples" [1]\nIntegrator "path" "string lightsamplestrategy" "uniform"\nWorldBegin\n']
    for i in range(0, len(recs), 24):
        r = recs[i]
        g, L = r["geom"], fl(r["L"])
        two = '"bool twosided" "%s"' % ("true" if r["two_sided"] else "false")
        if r["kind"] == 0:
            out.append('AttributeBegin\nAreaLightSource "diffuse" "rgb L" [%s] %s\nShape "trianglemesh" "integer indices" [0 1 2] "point P" [%s]\nAttributeEnd\n' % (L, two, fl(g[:9])))
        elif r["kind"] == 1:
            out.append('AttributeBegin\nTranslate %s\nAreaLightSource "diffuse" "rgb L" [%s] %s\nShape "sphere" "float radius" [%s]\nAttributeEnd\n' % (fl(g[:3]), L, two, fl(g[3:4])))
        elif r["kind"] == 2:
            out.append('LightSource "point" "rgb I" [%s] "point from" [%s]\n' % (L, fl(g[:3])))
        else:
            out.append('LightSource "spot" "rgb I" [%s] "point from" [%s] "point to" [%s] "float coneangle" [%s] "float conedeltaangle" [%s]\n' % (L, fl(g[:3]), fl(g[3:6]), fl(g[6:7]), fl(g[7:8])))
    out.append("WorldEnd\n")
    return "".join(out)


def scene_light_kat_scene(recs):
    """the one-triangle scene whose light i is the light of records [64 i, 64 i + 64) of light_vectors.npz 'scene_lights': distant lights, constant
    infinite lights and infinite lights with the radiance map scenes/envmap_40x20.pfm under Rotate a1 (x) . Rotate a2 (z)"""
    import numpy as np

    def fl(v):
        return " ".join("%.9g" % x for x in np.atleast_1d(v))
    out = ['Film "image" "integer xresolution" [8] "integer yresolution" [8] "string filename" "l.pfm"\nSampler "sobol" "integer pixelsamples" [1]\nIntegrator "path" "string lightsamplestrategy" "uniform"\nWorldBegin\n']
    for i in range(0, len(recs), 64):
        r = recs[i]
        g, L, kind = r["geom"], fl(r["L"]), int(r["kind"])
        if kind == 4:
            out.append('LightSource "distant" "rgb L" [%s] "point from" [%s] "point to" [%s]\n' % (L, fl(g[:3]), fl(g[3:6])))
        else:
            m = ' "string mapname" "%s"' % os.path.join(ROOT, "scenes", "envmap_40x20.pfm") if kind == 6 else ""
            out.append('AttributeBegin\nRotate %s 1 0 0\nRotate %s 0 0 1\nLightSource "infinite" "rgb L" [%s]%s\nAttributeEnd\n' % (fl(g[0]), fl(g[1]), L, m))
    out.append('Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]\nWorldEnd\n')
    return "".join(out)


def camera_kat_scene(r):
    """the scene of one camera configuration of tests/golden/camera_vectors.npz (record r): LookAt, perspective camera (pinhole / thin lens, optional
    frame aspect ratio), film resolution + crop window, Sobol' sampler"""
    import numpy as np

    def fl(v):
        return " ".join("%.9g" % x for x in np.atleast_1d(v))
    asp = ' "float frameaspectratio" [%s]' % fl(r["aspect"]) if r["aspect"] > 0 else ""
    return ('LookAt %s %s %s\nCamera "perspective" "float fov" [%s] "float lensradius" [%s] "float focaldistance" [%s]%s\n'
            'Sampler "sobol" "integer pixelsamples" [%d]\nPixelFilter "box"\n'
            'Film "image" "integer xresolution" [%d] "integer yresolution" [%d] "float cropwindow" [%s] "string filename" "c.pfm"\n'
            'WorldBegin\nLightSource "point" "rgb I" [1 1 1]\nShape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]\nWorldEnd\n'
            % (fl(r["eye"]), fl(r["look"]), fl(r["up"]), fl(r["fov"]), fl(r["lensr"]), fl(r["focald"]), asp, r["spp"], r["xres"], r["yres"], fl(r["crop"])))
