#define GPAMD_KIND gpamd::KIND_RBF
#define GPAMD_NAME rbf
#include "kvd_family.inc"
