// env_gate.h -- the one place where the library reads PIRGPU_* environment variables.
#pragma once
// Behaviour switches can be given by name (pirgpu_set_option) or -- for the A/B scripts under tools/ and the tests --
// through PIRGPU_<NAME> environment variables.  The environment is honoured ONLY when PIRGPU_ALLOW_ENV=1 is set as well:
// a server's arithmetic flavour or launch geometry does not change because of what happens to be in the environment of
// whoever starts it.  Host-side helper; returns nullptr when the variable is unset or the
// gate is closed.
#include <stdlib.h>
#include <string.h>
inline const char* pirgpu_env(const char* name) {
  const char* a = getenv("PIRGPU_ALLOW_ENV");
  return a && a[0] == '1' ? getenv(name) : nullptr;
}

