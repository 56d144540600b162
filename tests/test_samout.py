"""Output half of SURVEY.md §8 f-4: read tags of an unaligned record, pinned on the reference's golden values
for mean_qscore_from_qstring (tests/SequenceUtilsTest.cpp:105-139)."""
import numpy as np

from dorado_amd import samout


def test_mean_qscore_golden_values():
    assert samout.mean_qscore_from_qstring("") == 0.0
    rng = np.random.default_rng(42)
    for q in range(1, 51):
        s = chr(33 + q) * int(rng.integers(1, 101))
        assert abs(samout.mean_qscore_from_qstring(s) - q) <= 1e-4 * q
    assert samout.mean_qscore_from_qstring("!") == 1.0
    assert samout.mean_qscore_from_qstring("Z") == 50.0
    for s, want in [("$$$$$%$###%&$%$$$#$$%&//*.,+((())*((&&'&$$%/.)((-3:>1(-(4NB;?C@>78?B@3", 6.27468),
                    ("464887/55.519;@=>?0..,-./*)+$&&/00)*++-//-20?@===@D:9/=<:<E@AB;98(&$%&+*", 11.61238),
                    ("33B<87ESEA41GDDSGHDC?=>:84:<?568@", 23.70278),
                    ("%$$')*(,*+78665;3378H@=>A42004.", 10.62169)]:
        assert abs(samout.mean_qscore_from_qstring(s) - want) <= 1e-5 * want + 1e-5
    assert abs(samout.mean_qscore_from_qstring("####%%%%") - 2.88587) < 1e-4      # start position 0
    assert abs(samout.mean_qscore_from_qstring("####%%%%"[4:]) - 4.0) < 1e-4     # start position 4
    assert samout.calculate_mean_qscore("####%%%%", 60) == samout.mean_qscore_from_qstring("####%%%%")
    assert samout.calculate_mean_qscore("#" * 60 + "%%%%", 60) == samout.mean_qscore_from_qstring("%%%%")


def test_sam_record_layout():
    line = samout.sam_record("id1", "ACGT", "%%%%", moves=[1, 0, 1, 1, 0, 1], model_stride=6, num_samples=26,
                             num_trimmed_samples=10, sample_rate=4000, mux=2, channel=7,
                             start_time=samout.timestamp_from_unix_ms(1505209812456), read_number=12,
                             filename="a.pod5", shift_pa=93.5, scale_pa=23.25)
    f = line.split("\t")
    assert f[:11] == ["id1", "4", "*", "0", "0", "*", "*", "0", "0", "ACGT", "%%%%"]
    tags = dict((t[:2], t) for t in f[11:])
    assert [t[:2] for t in f[11:]] == ["qs", "du", "ns", "ts", "mx", "ch", "st", "rn", "fn", "sm", "sd", "sv", "dx", "mv"]
    assert tags["ns"] == "ns:i:36" and tags["ts"] == "ts:i:10" and tags["du"] == "du:f:0.009"
    assert tags["st"] == "st:Z:2017-09-12T09:50:12.456+00:00"
    assert tags["mv"] == "mv:B:c,6,1,0,1,1,0,1" and tags["sm"] == "sm:f:93.5" and tags["sv"] == "sv:Z:pa"
