#define GPAMD_KIND gpamd::KIND_RBF
#define GPAMD_NAME rbf
#include "kvm_family.inc"
