// shortrange_table.cpp -- host builder of the TreePM transition tables.
//
// Replaces (reference): the tabulation block of force_treeallocate (forcetree.c:3246-3403) with
// performConvolution / ngravsConvolutionInit (ngravs_core.c:72-184).  The reference samples the
// normalised k-space Green's function times exp(-k^2/4) at k_j = j*dk, takes a length-589 682
// backward DFT (x-spacing 1/32768), reads temp(u) at the bin centres and obtains tempI(u) by a
// running Newton-Cotes 3/8 integral of the transform.
//
// The DFT of that real symmetric sequence is the cosine series
//     temp(x)  = dk [ f(0) + 2 sum_j f(k_j) cos(k_j x) ]
// whose integral from 0 is known in closed form,
//     tempI(x) = dk [ f(0) x + 2 sum_j f(k_j) sin(k_j x) / k_j ],
// and f underflows to exactly 0 beyond k ~ 55 (j ~ 157), so both are evaluated directly at the
// NTAB bin centres: no 589 682-point transform (its length has the prime factor 9511), no
// quadrature error, same sampling -> same tables to rounding (tests/test_table.py).
#include <cmath>
#include <vector>
#include "engine.hpp"

double cfg_asmth(const ngravs_config_t *c)
{
  return c->asmth > 0 ? c->asmth : NGRAVS_ASMTH * c->box_size / c->pmgrid;   // pm_periodic.c:59
}
double cfg_rcut(const ngravs_config_t *c)
{
  return c->rcut > 0 ? c->rcut : NGRAVS_RCUT * cfg_asmth(c);                 // pm_periodic.c:60
}

// NormedGreensFxns: k^2 * G(k) in table units (ngravs.c:400, :880-885, :834; gridKtoNormK ngravs_core.c:27-35)
static double normed_green(const ngravs_config_t *c, double asmth, int law, double k2)
{
  double ym;
  switch(law)
    {
    case NGRAVS_LAW_NEWTON:
      return 1.0;
    case NGRAVS_LAW_NEG_NEWTON:
      return -1.0;
    case NGRAVS_LAW_YUKAWA:
    case NGRAVS_LAW_COLOYUK:
      ym = 4 * M_PI * asmth * (c->yukawa_imass / (2 * M_PI)) / c->box_size;
      return k2 / (k2 + ym * ym) * std::exp(-ym * ym * 0.25) + (law == NGRAVS_LAW_COLOYUK ? 1.0 : 0.0);
    default:
      return 0.0;
    }
}

void host_shortrange_table(const ngravs_config_t *cfg, double *force, double *pot)
{
  const int ntab = NTAB, len = 3, ol = 8;
  const double n = 12.0 * ntab * ol * len - 6.0 * ol * len + 2.0;          // ngravs_core.c:177
  const double dk = 2.0 * M_PI * ntab * 6.0 * ol / (3.0 * n);              // jTok(1)
  const double Z = 0.5;                                                    // forcetree.c:3276
  const double asmth = cfg->pmgrid > 0 ? cfg_asmth(cfg) : 0.0;
  const int ng = cfg->n_gravs;
  for(int nA = 0; nA < ng; nA++)     // sources
    for(int nB = 0; nB < ng; nB++)   // receivers; table[nB][nA] <- NormedGreensFxns[nB][nA] (forcetree.c:3281-3292)
      {
        int law = cfg->law_normed[nB][nA];
        std::vector<double> f, k;
        for(int j = 0; j < (int)(n / 2); j++)
          {
            double kj = dk * j, k2 = kj * kj;
            double v = normed_green(cfg, asmth, law, k2) * std::exp(-k2 * Z * Z);
            if(v == 0.0 && kj > 60.0)
              break;
            f.push_back(v);
            k.push_back(kj);
          }
        for(int i = 0; i < ntab; i++)
          {
            double u = 3.0 / ntab * (i + 0.5);
            double t = f[0], ti = f[0] * u;
            for(size_t j = 1; j < f.size(); j++)
              {
                t += 2.0 * f[j] * std::cos(k[j] * u);
                ti += 2.0 * f[j] * std::sin(k[j] * u) / k[j];
              }
            t *= dk;
            ti *= dk;
            ti /= u * u;                                                     // forcetree.c:3340-3354
            t /= u;
            if(pot)
              pot[((size_t)nB * ng + nA) * ntab + i] = t;
            force[((size_t)nB * ng + nA) * ntab + i] = ti - t;
          }
      }
}
