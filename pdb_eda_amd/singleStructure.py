"""Single-structure result tables and writers (ref pdb_eda/singleStructure.py:57-196).

The reference's ``pdb_eda single`` is a docopt CLI around one ``DensityAnalysis`` object; the CLI
is out of scope here (docopt / jsonpickle are not available), the *results* are not: ``TABLES`` maps every sub-mode to the header
and the ``DensityAnalysis`` call the reference's ``main`` uses for it (singleStructure.py:98-163), ``rows`` looks the
sub-mode up, and ``write`` emits the reference's JSON / CSV text (169-178), so
outputs diff cleanly against reference runs made elsewhere.  All numbers come from the MI355X path
behind ``DensityAnalysis``.
"""
import json
import sys

import numpy

from . import densityAnalysis

MODES = ("cloud", "density", "difference", "blob", "statistics")


def numpyConverter(obj):
    """Values json.dumps cannot write (numpy scalars / arrays) as plain Python ones (role of singleStructure.py:180-196)."""
    if isinstance(obj, numpy.generic):
        return obj.item()
    if isinstance(obj, numpy.ndarray):
        return [numpyConverter(item) for item in obj]
    return obj


def _plainColumns(table, listColumns=(), floatColumns=()):
    """Tuples / arrays in the given columns become lists (of floats) so every row serialises (singleStructure.py:119-121 ...)."""
    for row in table:
        for c in listColumns:
            row[c] = list(row[c])
        for c in floatColumns:
            row[c] = [float(v) for v in row[c]]
    return table


def _blobTable(analyzer, o):
    """blob sub-mode: statistics of the green and / or red Fo-Fc blobs, else of the blue 2Fo-Fc blobs (singleStructure.py:128-141)."""
    diffObj, densObj, numSD = analyzer.diffDensityObj, analyzer.densityObj, o["numSD"]
    if o["green"] and o["red"]:          # one fused pass over the Fo-Fc grid gives both lists
        lists = diffObj.createFullBlobLists(diffObj.meanDensity + numSD * diffObj.stdDensity)
    elif o["green"]:
        lists = [diffObj.createFullBlobList(diffObj.meanDensity + numSD * diffObj.stdDensity)]
    elif o["red"]:
        lists = [diffObj.createFullBlobList(-1 * (diffObj.meanDensity + numSD * diffObj.stdDensity))]
    else:
        lists = [densObj.createFullBlobList(densObj.meanDensity + numSD * densObj.stdDensity)]
    table = [row for blobs in lists for row in analyzer.calculateAtomSpecificBlobStatistics(blobs)]
    return _plainColumns(table, listColumns=(9,), floatColumns=(10, 11))


def _cloudTable(attribute):
    def table(analyzer, o):
        return [[numpyConverter(v) for v in item] + [analyzer.densityElectronRatio] for item in getattr(analyzer, attribute)]
    return table


_DA = densityAnalysis.DensityAnalysis
# Columns main() turns into lists for JSON: 4 and 5 = symmetry tag and coordinate of the atom-metrics rows.  The reference
# applies the SAME indices to the symmetry-atom density / discrepancy rows (singleStructure.py:119-121, 133-135), whose
# leading 'model' column shifts them onto atom_name (-> list of characters) and symmetry (-> floats): reproduced as is (Q11).
_SYM = dict(listColumns=(4,), floatColumns=(5,))
# (mode, level) -> (header of the table, rows of the table): one entry per sub-mode of `pdb_eda single` (singleStructure.py:97-163)
TABLES = {
    ("cloud", "atom"): (lambda an: list(map(str, an.atomCloudDescriptions.dtype.names)) + ['density_electron_ratio'], _cloudTable("atomCloudDescriptions")),
    ("cloud", "residue"): (lambda an: _DA.residueCloudHeader + ['density_electron_ratio'], _cloudTable("residueCloudDescriptions")),
    ("cloud", "domain"): (lambda an: _DA.domainCloudHeader + ['density_electron_ratio'], _cloudTable("domainCloudDescriptions")),
    ("density", "atom"): (lambda an: _DA.atomRegionDensityHeader,
                          lambda an, o: an.calculateAtomRegionDensity(o["radius"], o["numSD"], o["type"], o["optimizedRadii"])),
    ("density", "residue"): (lambda an: _DA.residueRegionDensityHeader,
                             lambda an, o: an.calculateResidueRegionDensity(o["radius"], o["numSD"], o["type"], o["atomMask"], o["optimizedRadii"])),
    ("density", "symmetry-atom"): (lambda an: _DA.symmetryAtomRegionDensityHeader,
                                   lambda an, o: _plainColumns(an.calculateSymmetryAtomRegionDensity(o["radius"], o["numSD"], o["type"], o["optimizedRadii"]), **_SYM)),
    ("difference", "atom"): (lambda an: _DA.atomRegionDiscrepancyHeader, lambda an, o: an.calculateAtomRegionDiscrepancies(o["radius"], o["numSD"], o["type"])),
    ("difference", "residue"): (lambda an: _DA.residueRegionDiscrepancyHeader,
                                lambda an, o: an.calculateResidueRegionDiscrepancies(o["radius"], o["numSD"], o["type"], o["atomMask"])),
    ("difference", "symmetry-atom"): (lambda an: _DA.symmetryAtomRegionDiscrepancyHeader,
                                      lambda an, o: _plainColumns(an.calculateSymmetryAtomRegionDiscrepancies(o["radius"], o["numSD"], o["type"]), **_SYM)),
    ("blob", None): (lambda an: _DA.blobStatisticsHeader, _blobTable),
    ("statistics", "residue"): (lambda an: an.residueMetricsHeaderList, lambda an, o: an.residueMetrics()),
    ("statistics", "atom"): (lambda an: an.atomMetricsHeaderList, lambda an, o: _plainColumns(an.atomMetrics(), **_SYM)),
}


def rows(analyzer, mode, level="atom", radius=3.5, numSD=None, type="", atomMask=None, optimizedRadii=False, green=False, red=False,
         includePdbid=False):
    """(headerList, rowList) of one ``pdb_eda single`` sub-mode, looked up in ``TABLES``.

    mode: cloud | density | difference | blob | statistics;  level: atom | residue | domain | symmetry-atom
    (the reference's --atom / --residue / --domain / --symmetry-atom; ignored by blob);  green / red: blob colours
    (neither = blue);  numSD default 3.0 for green / red / difference, else 1.5 (singleStructure.py:65-67)."""
    if mode not in MODES:
        raise ValueError("mode must be one of %s" % (MODES,))
    key = (mode, None if mode == "blob" else level)
    if key not in TABLES:
        raise ValueError("%s mode has the levels %s" % (mode, ", ".join(lv for md, lv in TABLES if md == mode and lv)))
    options = {"radius": float(radius), "numSD": float(numSD if numSD is not None else (3.0 if green or red or mode == "difference" else 1.5)),
               "type": type, "atomMask": atomMask, "optimizedRadii": optimizedRadii, "green": green, "red": red}
    if mode == "cloud":
        analyzer.aggregateCloud()
    header, table = TABLES[key]
    headerList, result = list(header(analyzer)), table(analyzer, options)
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
