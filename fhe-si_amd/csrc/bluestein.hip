// bluestein.hip -- general-m (non power-of-two) transforms: placeholder until the device Bluestein path lands.
#include "fhesi_internal.h"
int bluestein_init(fhesi_ctx* ctx) { (void)ctx; FHESI_FAIL("general (non power-of-two) m is not supported by this build yet"); }
void bluestein_destroy(fhesi_ctx* ctx) { (void)ctx; }
int launch_bluestein_fwd(fhesi_ctx*, u64*, i64, int, const int*) { FHESI_FAIL("general m not supported yet"); }
int launch_bluestein_inv(fhesi_ctx*, u64*, i64, int, const int*) { FHESI_FAIL("general m not supported yet"); }
