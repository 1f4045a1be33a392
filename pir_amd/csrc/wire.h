// wire.h -- pir/proto wire format + SEAL 3.5.6 object codec (host side).
#pragma once
#include <stddef.h>
#include <stdint.h>
