/* tests/glue_stub/proto.h -- TEST-ONLY prototypes (reference proto.h:34,36,61,77,78,83,84,86,88,89,96,108,113,114,149,150,155,161,163,180,184,190) */
#ifndef PROTO_H
#define PROTO_H
#include "allvars.h"
void do_box_wrapping(void);
void domain_Decomposition(void);
void force_treeallocate(int maxnodes, int maxpart);
int force_treebuild(int npart);
void force_treefree(void);
void force_treeevaluate_potential(int target, int mode);
void force_treeevaluate_potential_shortrange(int target, int mode);
void force_update_hmax(void);
void force_update_len(void);
void lattice_init(void);
void pmpotential_periodic(void);
void set_softenings(void);
void force_update_pseudoparticles(void);
double get_random_number(int id);
void gravity_forcetest(void);
void gravity_tree(void);
peanokey peano_hilbert_key(int x, int y, int z, int bits);
void peano_hilbert_order(void);
void pm_init_periodic(void);
void pmforce_periodic(void);
double second(void);
double timediff(double t0, double t1);
void endrun(int);
#endif
