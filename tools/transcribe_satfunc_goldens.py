#!/usr/bin/env python3
"""Transcribes the known-answer arrays (DATA only) of the reference's tests/test_satfunc.cpp into
tests/golden/satfunc_eps.json.  Run in the build container where /root/reference exists; the JSON is committed."""
import json, re, sys
src = open("/root/reference/tests/test_satfunc.cpp").read()
cases = {}
names = re.findall(r"BOOST_AUTO_TEST_CASE \((\w+)\)", src)
bodies = re.split(r"BOOST_AUTO_TEST_CASE \(\w+\)", src)[1:]
for name, body in zip(names, bodies):
    arrs = {}
    for m in re.finditer(r"double (krw|kro|DkrwDsw|DkroDsw|DkroDsg)(\[\w+\])+\s*=\s*(\{.*?\});", body, re.S):
        txt = m.group(3).replace("{", "[").replace("}", "]")
        arrs[m.group(1)] = json.loads(re.sub(r",\s*\]", "]", txt))
    tol = re.search(r"const double reltol = ([0-9.e+-]+);", body)
    deck = re.search(r'parseFile\("(\w+\.DATA)"', body)
    cases[name] = {"deck": deck.group(1) if deck else None, "reltol_percent": float(tol.group(1)), **arrs}
# per-cell scaled end points of the EPS decks (PROPS section; SWL/SWCR/SWU, everything else defaulted to the table values)
cases["GwsegEPS_A"]["endpoints"] = {"SWL": [0.1] * 4 + [0.2] * 4 + [0.1] * 2, "SWCR": [0.2, 0.2, 0.4, 0.4, 0.2, 0.2, 0.4, 0.4, 0.2, 0.2],
                                    "SWU": [0.9, 0.7, 0.9, 0.7, 0.9, 0.7, 0.9, 0.7, 0.9, 0.9]}
# EPS_C reaches the same values "the Norne way" (EQUALS / COPY / ADD / MULTIPLY): SWCR = SWL + add, SWU = 1 - SWL + add
swl = [0.1] * 4 + [0.2] * 4 + [0.1] * 2
cases["GwsegEPS_C"]["endpoints"] = {"SWL": swl,
    "SWCR": [s + a for s, a in zip(swl, [0.1, 0.1, 0.3, 0.3, 0.0, 0.0, 0.2, 0.2, 0.1, 0.1])],
    "SWU": [(-s + 1.0) + a for s, a in zip(swl, [0.0, -0.2, 0.0, -0.2, 0.1, -0.1, 0.1, -0.1, 0.0, 0.0])]}
out = {"_source": "Known answers transcribed from the reference's tests/test_satfunc.cpp (cases GwsegEPSBase :140-225, GwsegEPS_A :227-379, "
                  "GwsegEPS_C :480-582, GwsegEPS_D :584-...; decks tests/satfuncEPS*.DATA share the SWOF/SGOF of satfuncStandard.DATA). "
                  "s_w = i*0.1, s_o = 1 - s_w, s_g = 0 in cell icell; checked there with CHECK(value, expected, reltol) in percent.",
       "cases": {k: v for k, v in cases.items() if k != "GwsegStandard"}}
json.dump(out, open("tests/golden/satfunc_eps.json", "w"), indent=1)
for k, v in out["cases"].items():
    print(k, v["deck"], v["reltol_percent"], {a: (len(v[a]), len(v[a][0]) if isinstance(v[a][0], list) else 1) for a in v if a not in ("deck", "reltol_percent", "endpoints")})
