/* tests/glue_stub/ngravs.h -- TEST-ONLY prototypes of the force-law plug-ins the glue names (reference ngravs.h, ngravs.c:344-886) */
#ifndef NGRAVS_H
#define NGRAVS_H
#define LAW(f) double f(double, double, double, double, long)
LAW(none); LAW(newtonian); LAW(neg_newtonian); LAW(plummer); LAW(neg_plummer); LAW(pgdelta); LAW(neg_pgdelta); LAW(normed_pgdelta);
LAW(bambam); LAW(sourcebambaryon); LAW(sourcebaryonbam); LAW(bambam_spline); LAW(sourcebambaryon_spline); LAW(sourcebaryonbam_spline);
LAW(yukawa); LAW(pgyukawa); LAW(normed_pgyukawa); LAW(coloyuk); LAW(pgcoloyuk); LAW(normed_pgcoloyuk);
#undef LAW
#endif
