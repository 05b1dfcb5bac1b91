#!/usr/bin/env perl
# run.pl -- local job launcher with the calling convention the recipes use for ${train_cmd} / ${cuda_cmd}
# (egs/*/cmd.sh: train_cmd="run.pl", cuda_cmd="run.pl --gpu 1"):
#
#     run.pl [--gpu N] [--num-threads N] [--max-jobs-run N] [--config FILE] [JOB=A:B] LOG COMMAND [ARGS...]
#
# Runs COMMAND through bash on this machine with stdout and stderr appended to LOG (the directory is created), a
# header naming the command and a footer with the elapsed time and exit status.  With JOB=A:B the command runs once
# per index, "JOB" replaced by the index in LOG and in the command, at most --max-jobs-run at a time.  Resource
# options (--gpu, --num-threads, --mem, --config) are what a cluster launcher would act on; here they are accepted
# and ignored except that --gpu 0 hides the GPUs from the job.  Exit status: 0 when every job succeeded, 1 otherwise.
# Written for this repository; the recipes find it on PATH through path.sh.
use strict;
use warnings;
use File::Basename;
use File::Path qw(make_path);

my ($gpu, $max_jobs, $jobname, $jobstart, $jobend) = (undef, 0, undef, undef, undef);
while (@ARGV) {
    my $a = $ARGV[0];
    if ($a =~ /^--(gpu|num-threads|max-jobs-run|config|mem)$/) {
        shift @ARGV;
        die "run.pl: option $a needs a value\n" unless @ARGV;
        my $v = shift @ARGV;
        $gpu = $v if $a eq "--gpu";
        $max_jobs = $v if $a eq "--max-jobs-run";
    } elsif ($a =~ /^([A-Za-z_][A-Za-z0-9_]*)=(\d+):(\d+)$/) {
        ($jobname, $jobstart, $jobend) = ($1, $2, $3);
        die "run.pl: empty job range $a\n" if $jobstart > $jobend;
        shift @ARGV;
    } elsif ($a =~ /^--/) {
        die "run.pl: unknown option $a\n";
    } else {
        last;
    }
}
die "usage: run.pl [options] [JOB=A:B] log-file command-line\n" if @ARGV < 2;
my $log = shift @ARGV;
# an argument with whitespace (or an empty one) is handed to the shell quoted; words without whitespace pass bare, so
# a command line written as separate words may contain shell operators ("a", "|", "b")
my $cmd = join(" ", map { $_ eq "" ? "''" : !/\s/ ? $_ : !/'/ ? "'$_'" : "\"$_\"" } @ARGV);
($jobstart, $jobend) = (1, 1) unless defined $jobname;
$max_jobs = $jobend - $jobstart + 1 if $max_jobs <= 0;

sub run_one {
    my ($idx) = @_;
    my ($l, $c) = ($log, $cmd);
    if (defined $jobname) {
        $l =~ s/$jobname/$idx/g;
        $c =~ s/$jobname/$idx/g;
    }
    make_path(dirname($l));
    open(my $fh, ">", $l) or die "run.pl: cannot write $l: $!\n";
    print $fh "# $c\n# Started at " . localtime() . "\n#\n";
    close($fh);
    my $t0 = time();
    $ENV{HIP_VISIBLE_DEVICES} = "" if defined $gpu && $gpu eq "0";
    my $status = system("bash", "-c", "( $c ) >> '$l' 2>&1");
    my $rc = $status == -1 ? 127 : ($status & 127 ? 128 + ($status & 127) : $status >> 8);
    open($fh, ">>", $l) or die "run.pl: cannot append to $l: $!\n";
    print $fh "# Accounting: time=" . (time() - $t0) . " threads=1\n";
    print $fh "# Finished at " . localtime() . " with status $rc\n";
    close($fh);
    return $rc;
}

my ($failed, $running, %pids) = (0, 0);
for my $idx ($jobstart .. $jobend) {
    if ($jobend == $jobstart) {
        $failed++ if run_one($idx) != 0;
        last;
    }
    while ($running >= $max_jobs) {
        my $p = wait();
        last if $p < 0;
        $failed++ if $? != 0;
        $running--;
    }
    my $pid = fork();
    die "run.pl: fork failed: $!\n" unless defined $pid;
    if ($pid == 0) { exit(run_one($idx) == 0 ? 0 : 1); }
    $running++;
}
while ($running > 0) {
    my $p = wait();
    last if $p < 0;
    $failed++ if $? != 0;
    $running--;
}
if ($failed) {
    my $n = $jobend - $jobstart + 1;
    print STDERR "run.pl: $failed / $n failed, log is in $log\n";
    exit 1;
}
exit 0;
