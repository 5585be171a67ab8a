// Version of the Spectra API these headers follow (the reference is v1.2.0, Util/Version.h:10-14) and of this
// implementation.
#ifndef MISPEC_SPECTRA_VERSION_H
#define MISPEC_SPECTRA_VERSION_H

#define SPECTRA_MAJOR_VERSION 1
#define SPECTRA_MINOR_VERSION 2
#define SPECTRA_PATCH_VERSION 0
#define SPECTRA_VERSION (SPECTRA_MAJOR_VERSION * 10000 + SPECTRA_MINOR_VERSION * 100 + SPECTRA_PATCH_VERSION)

// this implementation (mispec_version() of the shared library reports the same numbers)
#define MISPEC_MAJOR_VERSION 0
#define MISPEC_MINOR_VERSION 1
#define MISPEC_PATCH_VERSION 0

#endif
