"""Golden vectors for optimise mode (BASELINE configs[4]) and for every `pdb_eda single` table: the REFERENCE's
DensityAnalysis run on the two synthetic analysis entries (make_golden_analysis.entry)

  * once per parameter set of pdb_eda_amd.synthetic.sweep_param_sets() -- the per-entry record optimizeParams.processFunction
    keeps (optimizeParams.py:428-436: diffs, slopes, overlap counters), built from the analyzer's attributes exactly as
    those lines do (processFunction itself needs docopt + a download and cannot be imported);
  * once with the base parameters for the method outputs behind each sub-mode of singleStructure.main
    (singleStructure.py:97-163), stored as generic JSON.

Build container only:  python tests/golden/make_golden_sweep.py      -> tests/golden/analysis_sweep.npz
Inputs are regenerated from seeds by the test (not stored again); outputs are numbers and names, never reference source.
"""
import io
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refload  # noqa: E402
import make_golden_analysis as mga  # noqa: E402
from pdb_eda_amd import synthetic  # noqa: E402
from pdb_eda_amd import structure as my_structure  # noqa: E402


def plain(obj):
    """numpy scalars / arrays / tuples / sets -> JSON types (what singleStructure.numpyConverter + json.dumps do)."""
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.floating,)):
        return float(obj)
    if isinstance(obj, (np.bool_,)):
        return bool(obj)
    if isinstance(obj, np.ndarray):
        return [plain(x) for x in obj.tolist()]
    if isinstance(obj, (list, tuple)):
        return [plain(x) for x in obj]
    if isinstance(obj, dict):
        return {str(k): plain(v) for k, v in obj.items()}
    if isinstance(obj, np.void):
        return [plain(x) for x in obj]
    return obj


def analyzer_for(ccp4, da, name, params):
    spec, st, _, dens, diff, rot = mga.entry(name)
    da.setGlobals(params)
    densityObj = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, dens)), name)
    diffObj = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, diff)), name)
    densityObj.densityCutoff = densityObj.meanDensity + 1.5 * densityObj.stdDensity
    diffObj.diffDensityCutoff = diffObj.meanDensity + 3 * diffObj.stdDensity
    pdbObj = my_structure.PDBEntry(my_structure.PDBHeader(pdbid=name, resolution=2.0, spaceGroup="P_1", rotationMats=rot))
    st.header = {"resolution": 2.0}
    return da.DensityAnalysis(name, densityObj, diffObj, st, pdbObj)


def main():
    ccp4, da = refload.load()
    out = {}
    sets = synthetic.sweep_param_sets()
    for name in mga.CASES:
        for k, params in enumerate(sets):
            an = analyzer_for(ccp4, da, name, params)
            ratio = an.densityElectronRatio
            assert ratio, (name, k)
            med = an.medians['corrected_density_electron_ratio']
            # optimizeParams.py:428-436
            diffs = {t: ((med[t] - ratio) / ratio) for t in params["radii"] if t in med and not np.isnan(med[t])}
            slopes = {t: an.medians['slopes'][t] for t in params["slopes"] if t in an.medians['slopes'] and not np.isnan(an.medians['slopes'][t])}
            out["%s_k%d" % (name, k)] = np.array(json.dumps(plain({
                "ratio": ratio, "diffs": diffs, "slopes": slopes, "num_voxels": an.numVoxelsAggregated, "total_electrons": an.totalAggregatedElectrons,
                "atomtype_overlap_completeness": dict(an.atomTypeOverlapCompleteness),
                "atomtype_overlap_incompleteness": dict(an.atomTypeOverlapIncompleteness), "num_atoms": len(an.atomCloudDescriptions)})))
            print(name, k, ratio, flush=True)
        # ---- the `pdb_eda single` tables with the base parameters and the CLI's defaults (radius 3.5; numSD 3.0 / 1.5) ----
        an = analyzer_for(ccp4, da, name, sets[0])
        an.aggregateCloud()
        t = {}
        t["cloud/atom"] = [list(item) for item in an.atomCloudDescriptions]
        t["cloud/atom/names"] = list(an.atomCloudDescriptions.dtype.names)
        t["cloud/residue"] = an.residueCloudDescriptions
        t["cloud/domain"] = an.domainCloudDescriptions
        t["density/atom"] = an.calculateAtomRegionDensity(3.5, 1.5, "", False)
        t["density/residue"] = an.calculateResidueRegionDensity(3.5, 1.5, "", None, False)
        t["density/symmetry-atom"] = an.calculateSymmetryAtomRegionDensity(3.5, 1.5, "", False)
        print(name, "density done", flush=True)
        t["difference/atom"] = an.calculateAtomRegionDiscrepancies(3.5, 3.0, "")
        t["difference/residue"] = an.calculateResidueRegionDiscrepancies(3.5, 3.0, "", None)
        t["difference/symmetry-atom"] = an.calculateSymmetryAtomRegionDiscrepancies(3.5, 3.0, "")
        print(name, "difference done", flush=True)
        diffObj, densObj = an.diffDensityObj, an.densityObj
        t["blob/green"] = an.calculateAtomSpecificBlobStatistics(diffObj.createFullBlobList(diffObj.meanDensity + 3.0 * diffObj.stdDensity))
        t["blob/red"] = an.calculateAtomSpecificBlobStatistics(diffObj.createFullBlobList(-1 * (diffObj.meanDensity + 3.0 * diffObj.stdDensity)))
        t["blob/blue"] = an.calculateAtomSpecificBlobStatistics(densObj.createFullBlobList(densObj.meanDensity + 1.5 * densObj.stdDensity))
        print(name, "blob done", flush=True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t["statistics/residue"] = an.residueMetrics()
            t["statistics/atom"] = an.atomMetrics()
        t["ratio"] = an.densityElectronRatio
        out["%s_tables" % name] = np.array(json.dumps(plain(t)))
    path = os.path.join(HERE, "analysis_sweep.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
