/* Test double, see Rinternals.h in this directory. */
#ifndef FLGP_R_MOCK_R_H
#define FLGP_R_MOCK_R_H
#endif
