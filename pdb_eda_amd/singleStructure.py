"""Single-structure result tables and writers (ref pdb_eda/singleStructure.py:57-196).

The reference's ``pdb_eda single`` is a docopt CLI around one ``DensityAnalysis`` object; the CLI
is out of scope here (docopt / jsonpickle are not available), the *results* are not: ``rows`` builds the
header list + row list of every sub-mode exactly as the reference's ``main`` does
(singleStructure.py:98-163), and ``write`` emits the reference's JSON / CSV text (169-178), so
outputs diff cleanly against reference runs made elsewhere.  All numbers come from the MI355X path
behind ``DensityAnalysis``.
"""
import json
import sys

import numpy

from . import densityAnalysis

MODES = ("cloud", "density", "difference", "blob", "statistics")


def numpyConverter(obj):
    """ref singleStructure.py:180-196."""
    if isinstance(obj, numpy.integer):
        return int(obj)
    elif isinstance(obj, numpy.floating):
        return float(obj)
    elif isinstance(obj, numpy.ndarray):
        return [numpyConverter(item) for item in obj]
    return obj


def rows(analyzer, mode, level="atom", radius=3.5, numSD=None, type="", atomMask=None, optimizedRadii=False, green=False, red=False,
         includePdbid=False):
    """(headerList, rowList) of one ``pdb_eda single`` sub-mode.

    mode: cloud | density | difference | blob | statistics;  level: atom | residue | domain | symmetry-atom
    (the reference's --atom / --residue / --domain / --symmetry-atom);  green / red: blob colours
    (neither = blue);  numSD default 3.0 for green / red / difference, else 1.5 (singleStructure.py:65-67)."""
    DA = densityAnalysis.DensityAnalysis
    if numSD is None:
        numSD = 3.0 if green or red or mode == "difference" else 1.5
    numSD = float(numSD)
    radius = float(radius)
    if mode == "cloud":
        analyzer.aggregateCloud()
        ratio = analyzer.densityElectronRatio
        if level == "atom":
            headerList = list(map(str, list(analyzer.atomCloudDescriptions.dtype.names) + ['density_electron_ratio']))
            result = [[numpyConverter(element) for element in item] + [ratio] for item in analyzer.atomCloudDescriptions]
        elif level == "residue":
            headerList = DA.residueCloudHeader + ['density_electron_ratio']
            result = [list(item) + [ratio] for item in analyzer.residueCloudDescriptions]
        elif level == "domain":
            headerList = DA.domainCloudHeader + ['density_electron_ratio']
            result = [list(item) + [ratio] for item in analyzer.domainCloudDescriptions]
        else:
            raise ValueError("cloud mode has atom, residue and domain levels")
    elif mode == "density":
        if level == "atom":
            headerList = DA.atomRegionDensityHeader
            result = analyzer.calculateAtomRegionDensity(radius, numSD, type, optimizedRadii)
        elif level == "residue":
            headerList = DA.residueRegionDensityHeader
            result = analyzer.calculateResidueRegionDensity(radius, numSD, type, atomMask, optimizedRadii)
        elif level == "symmetry-atom":
            headerList = DA.symmetryAtomRegionDensityHeader
            result = analyzer.calculateSymmetryAtomRegionDensity(radius, numSD, type, optimizedRadii)
            for atomInfo in result:
                atomInfo[4] = [val for val in atomInfo[4]]
                atomInfo[5] = [float(val) for val in atomInfo[5]]
        else:
            raise ValueError("density mode has atom, residue and symmetry-atom levels")
    elif mode == "difference":
        if level == "atom":
            headerList = DA.atomRegionDiscrepancyHeader
            result = analyzer.calculateAtomRegionDiscrepancies(radius, numSD, type)
        elif level == "residue":
            headerList = DA.residueRegionDiscrepancyHeader
            result = analyzer.calculateResidueRegionDiscrepancies(radius, numSD, type, atomMask)
        elif level == "symmetry-atom":
            headerList = DA.symmetryAtomRegionDiscrepancyHeader
            result = analyzer.calculateSymmetryAtomRegionDiscrepancies(radius, numSD, type)
            for atomInfo in result:
                atomInfo[4] = [val for val in atomInfo[4]]
                atomInfo[5] = [float(val) for val in atomInfo[5]]
        else:
            raise ValueError("difference mode has atom, residue and symmetry-atom levels")
    elif mode == "blob":
        headerList = DA.blobStatisticsHeader
        result = []
        diffObj, densObj = analyzer.diffDensityObj, analyzer.densityObj
        if green and red:     # one fused pass over the Fo-Fc grid gives both lists
            g, r = diffObj.createFullBlobLists(diffObj.meanDensity + numSD * diffObj.stdDensity)
            result.extend(analyzer.calculateAtomSpecificBlobStatistics(g))
            result.extend(analyzer.calculateAtomSpecificBlobStatistics(r))
        elif green:
            result.extend(analyzer.calculateAtomSpecificBlobStatistics(diffObj.createFullBlobList(diffObj.meanDensity + numSD * diffObj.stdDensity)))
        elif red:
            result.extend(analyzer.calculateAtomSpecificBlobStatistics(diffObj.createFullBlobList(-1 * (diffObj.meanDensity + numSD * diffObj.stdDensity))))
        else:                 # blue by default
            result.extend(analyzer.calculateAtomSpecificBlobStatistics(densObj.createFullBlobList(densObj.meanDensity + numSD * densObj.stdDensity)))
        for blobInfo in result:
            blobInfo[9] = [val for val in blobInfo[9]]
            blobInfo[10] = [float(val) for val in blobInfo[10]]
            blobInfo[11] = [float(val) for val in blobInfo[11]]
    elif mode == "statistics":
        if level == "residue":
            headerList = analyzer.residueMetricsHeaderList
            result = analyzer.residueMetrics()
        elif level == "atom":
            headerList = analyzer.atomMetricsHeaderList
            result = analyzer.atomMetrics()
            for atomInfo in result:
                atomInfo[4] = [x for x in atomInfo[4]]
                atomInfo[5] = [float(x) for x in atomInfo[5]]
        else:
            raise ValueError("statistics mode has atom and residue levels")
    else:
        raise ValueError("mode must be one of %s" % (MODES,))
    headerList = list(headerList)
    if includePdbid:
        headerList = ["pdbid"] + headerList
        result = [[analyzer.pdbid] + list(row) for row in result]
    return headerList, result


def validationLine(analyzer):
    """The text ``pdb_eda single ... statistics --print-validation`` prints (singleStructure.py:146-148)."""
    medianAbsFo, medianAbsFc = analyzer.medianAbsFoFc()
    return "Median abs Fo(<1sd): %s Median abs Fc(<1sd): %s Relative Difference: %s" % (medianAbsFo, medianAbsFc, (medianAbsFo - medianAbsFc) / max(medianAbsFo, medianAbsFc))


def dumps(headerList, result, outFormat="json"):
    """The text the reference writes (singleStructure.py:169-178): CSV = header row + str() of every cell joined by
    commas; JSON = list of {header: value} objects, indent 2, sorted keys."""
    if outFormat == 'csv':
        return '\n'.join(','.join(map(str, row)) for row in [headerList] + list(result)) + '\n'
    return json.dumps([dict(zip(headerList, [numpyConverter(v) for v in row])) for row in result], indent=2, sort_keys=True) + '\n'


def write(headerList, result, outFile="-", outFormat="json"):
    text = dumps(headerList, result, outFormat)
    if outFile == "-":
        sys.stdout.write(text)
    else:
        with open(outFile, 'w') as fh:
            fh.write(text)
