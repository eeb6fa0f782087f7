// BSSRDFTable for the subsurface materials: what SubsurfaceMaterial / KdSubsurfaceMaterial compute in their
// constructors -- ComputeBeamDiffusionBSSRDF(g, eta, &table) with table(100, 64) (materials/subsurface.h:69-76, core/bssrdf.cpp:145-176):
// photon beam diffusion, single + multiple scattering, tabulated over 100 albedos x 64 optical radii, plus the effective albedo and the
// per-albedo CDF of the spline interpolant.  Float arithmetic in the reference's order (the table feeds a bit-exact oracle).
#include <cmath>
#include <map>
#include <mutex>

#include "scene.h"

namespace pbrt_amd {
namespace {
const Float kInv4Pi = 0.07957747154594766788;   // kPi: geom.h

// core/bssrdf.cpp:43-67 -- polynomial fits; one coefficient of the first is a double constant in the reference and is kept as one
Float FresnelMoment1(Float eta) {
    Float eta2 = eta * eta, eta3 = eta2 * eta, eta4 = eta3 * eta, eta5 = eta4 * eta;
    if (eta < 1) return 0.45966f - 1.73965f * eta + 3.37668f * eta2 - 3.904945 * eta3 + 2.49277f * eta4 - 0.68441f * eta5;
    return -4.61686f + 11.1136f * eta - 10.4646f * eta2 + 5.11455f * eta3 - 1.27198f * eta4 + 0.12746f * eta5;
}
Float FresnelMoment2(Float eta) {
    Float eta2 = eta * eta, eta3 = eta2 * eta, eta4 = eta3 * eta, eta5 = eta4 * eta;
    if (eta < 1) return 0.27614f - 0.87350f * eta + 1.12077f * eta2 - 0.65095f * eta3 + 0.07883f * eta4 + 0.04860f * eta5;
    Float r_eta = 1 / eta, r_eta2 = r_eta * r_eta, r_eta3 = r_eta2 * r_eta;
    return -547.033f + 45.3087f * r_eta3 - 218.725f * r_eta2 + 458.843f * r_eta + 404.557f * eta - 189.519f * eta2 + 54.9327f * eta3 - 9.00603f * eta4 +
           0.63942f * eta5;
}
Float FrDielectric(Float cosThetaI, Float etaI, Float etaT) {   // core/reflection.cpp:47-68
    cosThetaI = Clamp(cosThetaI, -1, 1);
    if (!(cosThetaI > 0.f)) { std::swap(etaI, etaT); cosThetaI = std::abs(cosThetaI); }
    Float sinThetaI = std::sqrt(std::max((Float)0, 1 - cosThetaI * cosThetaI));
    Float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1) return 1;
    Float cosThetaT = std::sqrt(std::max((Float)0, 1 - sinThetaT * sinThetaT));
    Float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    Float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}
Float PhaseHG(Float cosTheta, Float g) {   // core/medium.h:69-72
    Float denom = 1 + g * g + 2 * g * cosTheta;
    return kInv4Pi * (1 - g * g) / (denom * std::sqrt(denom));
}
// core/bssrdf.cpp:69-117: multiple scattering, 100 exponentially distributed real-source depths of the dipole
Float BeamDiffusionMS(Float sigma_s, Float sigma_a, Float g, Float eta, Float r) {
    const int nSamples = 100;
    Float Ed = 0;
    Float sigmap_s = sigma_s * (1 - g);
    Float sigmap_t = sigma_a + sigmap_s;
    Float rhop = sigmap_s / sigmap_t;
    Float D_g = (2 * sigma_a + sigmap_s) / (3 * sigmap_t * sigmap_t);
    Float sigma_tr = std::sqrt(sigma_a / D_g);
    Float fm1 = FresnelMoment1(eta), fm2 = FresnelMoment2(eta);
    Float ze = -2 * D_g * (1 + 3 * fm2) / (1 - 2 * fm1);
    Float cPhi = .25f * (1 - 2 * fm1), cE = .5f * (1 - 3 * fm2);
    for (int i = 0; i < nSamples; ++i) {
        Float zr = -std::log(1 - (i + .5f) / nSamples) / sigmap_t;
        Float zv = -zr + 2 * ze;
        Float dr = std::sqrt(r * r + zr * zr), dv = std::sqrt(r * r + zv * zv);
        Float phiD = kInv4Pi / D_g * (std::exp(-sigma_tr * dr) / dr - std::exp(-sigma_tr * dv) / dv);
        Float EDn = kInv4Pi * (zr * (1 + sigma_tr * dr) * std::exp(-sigma_tr * dr) / (dr * dr * dr) - zv * (1 + sigma_tr * dv) * std::exp(-sigma_tr * dv) / (dv * dv * dv));
        Float E = phiD * cPhi + EDn * cE;
        Float kappa = 1 - std::exp(-2 * sigmap_t * (dr + zr));
        Ed += kappa * rhop * rhop * E;
    }
    return Ed / nSamples;
}
// core/bssrdf.cpp:119-143: single scattering beyond the critical angle
Float BeamDiffusionSS(Float sigma_s, Float sigma_a, Float g, Float eta, Float r) {
    Float sigma_t = sigma_a + sigma_s, rho = sigma_s / sigma_t;
    Float tCrit = r * std::sqrt(eta * eta - 1);
    Float Ess = 0;
    const int nSamples = 100;
    for (int i = 0; i < nSamples; ++i) {
        Float ti = tCrit - std::log(1 - (i + .5f) / nSamples) / sigma_t;
        Float d = std::sqrt(r * r + ti * ti);
        Float cosThetaO = ti / d;
        Ess += rho * std::exp(-sigma_t * (d + tCrit)) / (d * d) * PhaseHG(cosThetaO, g) * (1 - FrDielectric(-cosThetaO, 1, eta)) * std::abs(cosThetaO);
    }
    return Ess / nSamples;
}
// IntegrateCatmullRom core/interpolation.cpp:260-286: running integral of the spline through (x, values)
Float IntegrateCatmullRom(int n, const Float *x, const Float *values, Float *cdf) {
    Float sum = 0;
    cdf[0] = 0;
    for (int i = 0; i < n - 1; ++i) {
        Float x0 = x[i], x1 = x[i + 1], f0 = values[i], f1 = values[i + 1], width = x1 - x0;
        Float d0 = i > 0 ? width * (f1 - values[i - 1]) / (x1 - x[i - 1]) : f1 - f0;
        Float d1 = i + 2 < n ? width * (values[i + 2] - f0) / (x[i + 2] - x0) : f1 - f0;
        sum += ((d0 - d1) * (1.f / 12.f) + (f0 + f1) * .5f) * width;
        cdf[i + 1] = sum;
    }
    return sum;
}
}  // namespace

std::shared_ptr<BSSRDFTableData> MakeBSSRDFTable(Float g, Float eta) {
    static std::mutex mu;
    static std::map<std::pair<Float, Float>, std::shared_ptr<BSSRDFTableData>> cache;   // a pure function of (g, eta): shared between materials
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_pair(g, eta);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    auto t = std::make_shared<BSSRDFTableData>();
    const int nRho = 100, nRadius = 64;
    t->nRho = nRho; t->nRadius = nRadius;
    t->rhoSamples.resize(nRho); t->radiusSamples.resize(nRadius); t->profile.resize(nRho * nRadius); t->rhoEff.resize(nRho); t->profileCDF.resize(nRho * nRadius);
    t->radiusSamples[0] = 0;
    t->radiusSamples[1] = 2.5e-3f;
    for (int i = 2; i < nRadius; ++i) t->radiusSamples[i] = t->radiusSamples[i - 1] * 1.2f;
    for (int i = 0; i < nRho; ++i) t->rhoSamples[i] = (1 - std::exp(-8 * i / (Float)(nRho - 1))) / (1 - std::exp(-8));
    for (int i = 0; i < nRho; ++i) {
        for (int j = 0; j < nRadius; ++j) {
            Float rho = t->rhoSamples[i], r = t->radiusSamples[j];
            t->profile[i * nRadius + j] = 2 * kPi * r * (BeamDiffusionSS(rho, 1 - rho, g, eta, r) + BeamDiffusionMS(rho, 1 - rho, g, eta, r));
        }
        t->rhoEff[i] = IntegrateCatmullRom(nRadius, t->radiusSamples.data(), &t->profile[i * nRadius], &t->profileCDF[i * nRadius]);
    }
    cache[key] = t;
    return t;
}

}  // namespace pbrt_amd
